"""Codec plugin surface -- same names and meaning as cbench/codecs/base.py:10-61."""
import abc
from typing import Any, Dict


class CodecInterface(abc.ABC):
    @abc.abstractmethod
    def compress(self, data, *args, **kwargs) -> bytes:
        pass

    @abc.abstractmethod
    def decompress(self, data: bytes, *args, **kwargs):
        pass

    def update_state(self, *args, **kwargs) -> None:  # optional: cache tables for faster coding
        pass


class VariableRateCodecInterface(abc.ABC):
    @abc.abstractmethod
    def set_rate_level(self, level, *args, **kwargs) -> None:
        pass

    @property
    def num_rate_levels(self) -> int:
        return 1


class VariableComplexityCodecInterface(abc.ABC):
    @abc.abstractmethod
    def set_complex_level(self, level, *args, **kwargs) -> None:
        pass

    def get_current_complex_metrics(self, *args, **kwargs) -> Dict[str, Any]:
        return dict()

    @property
    def num_complex_levels(self) -> int:
        return 1


class VariableTaskCodecInterface(abc.ABC):
    @abc.abstractmethod
    def set_task(self, task, *args, **kwargs) -> bool:
        pass

    @property
    def num_tasks(self) -> int:
        return 1
