"""Dataset config base (reference: src/dataset/util.py:9-16)."""
from abc import ABC, abstractmethod

import torch.utils.data as data
from pydantic import BaseModel


class DatasetConfig(BaseModel, ABC):
    batch_size: int = 32
    shuffle: bool = True
    num_workers: int = 8

    @abstractmethod
    def get_dataset(self) -> data.Dataset: ...
