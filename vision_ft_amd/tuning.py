"""Library-GEMM solution selection for gfx950.

The frozen-base projections (x·Wᵀ and dy·W of every nn.Linear the reference builds,
src/models/*/denoiser.py) stay on hipBLASLt / rocBLAS: they are plain GEMMs.  The libraries' default
heuristic is 2-10 % off their own best solution for the denoisers' shapes (e.g. 640x65536x640,
1280x16384x10240), so the solution index per (op, transposes, m, n, k, ld) is picked once by timing
all candidates on an MI355X (PyTorch's TunableOp) and the table is committed next to this file.
`enable()` loads that table and leaves tuning off: shapes that are not in it run the default solution,
nothing is timed at run time, and the calls stay HIP-graph capturable.

`enable(tune=True)` re-times shapes that are missing (first call of each shape, eager warm-up only)
and appends them to the table — how tuning/gfx950_gemm.csv was produced (tools/tune_gemms.sh).
"""
from __future__ import annotations

import os

import torch

DEFAULT_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning_tables", "gfx950_gemm.csv")


def enable(table: str | None = None, tune: bool = False, max_tuning_ms: int = 40) -> bool:
    """Select tuned library GEMM solutions from `table` (default: the committed gfx950 table).
    Returns False (and changes nothing) when the table does not exist and tune is False."""
    import torch.cuda.tunable as tunable

    table = table or DEFAULT_TABLE
    if not tune and not os.path.exists(table):
        return False
    tunable.enable(True)
    if not tune:
        # TunableOp may rewrite its file at process exit; with one process per GPU that would be N writers on the
        # committed table.  Lookup-only runs therefore work on a private copy.
        import shutil
        import tempfile

        private = os.path.join(tempfile.gettempdir(), f"vft_gemm_table_{os.getpid()}.csv")
        shutil.copyfile(table, private)
        table = private
    tunable.set_filename(table, insert_device_ordinal=False)
    tunable.tuning_enable(tune)
    if tune:
        tunable.set_max_tuning_duration(max_tuning_ms)
    return True


def disable() -> None:
    import torch.cuda.tunable as tunable

    tunable.enable(False)


def is_enabled() -> bool:
    return torch.cuda.tunable.is_enabled()
