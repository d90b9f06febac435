from typing import Literal

from pydantic import BaseModel

PEFT_TYPE = Literal["lora", "loha", "none"]


class PeftConfigMixin(BaseModel):
    """Base of adapter configs (reference: src/modules/peft/config.py:7-10)."""

    type: PEFT_TYPE
    dtype: str = "bfloat16"
