"""Module-name targeting for adapters/quantisation.

Same behaviour as src/utils/state_dict.py:9-42 of the reference: a plain string matches as a
substring, a ``RegexMatch`` matches with ``re.match`` (anchored at the start only); includes are
unioned first, then excludes removed.  The result order is unspecified (set) — callers treat it as
a set.
"""
from __future__ import annotations

import re
from typing import Iterable, Sequence

from pydantic import BaseModel


class RegexMatch(BaseModel):
    regex: str

    def __call__(self, value: str) -> bool:
        return re.match(self.regex, value) is not None


def _select(pattern: str | RegexMatch, keys: Iterable[str]) -> set[str]:
    if isinstance(pattern, RegexMatch):
        rx = re.compile(pattern.regex)
        return {k for k in keys if rx.match(k)}
    return {k for k in keys if pattern in k}


def get_target_keys(
    include: Sequence[str | RegexMatch],
    exclude: Sequence[str | RegexMatch],
    keys: list[str],
) -> list[str]:
    chosen: set[str] = set()
    for pat in include:
        chosen |= _select(pat, keys)
    for pat in exclude:
        chosen -= _select(pat, keys)
    return list(chosen)
