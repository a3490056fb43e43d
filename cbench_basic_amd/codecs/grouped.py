"""GroupedVariableRateCodec -- cbench/codecs/base.py:138-243: a bank of codecs (one per rate point, e.g. the four
lambda-codecs of configs/presets/lossy_latent_graph_scalable_ar_models.py:733-757) behind ONE codec interface.

``set_rate_level`` selects the active member (optionally through ``codec_vr_level_config``: outer level -> (member,
member's own rate level)); complexity levels and tasks fan out to every member unless ``active_only``; compress /
decompress / forward_estimate_bitlen go to the active member; update_state / post_training_process visit all.  Members are
registered as ``codec_{i}`` so a checkpoint of the reference's grouped codec (keys ``codec_0.entropy_coder...``) loads."""
from typing import Any, Dict, List, Tuple

from ..base import HotPathModule
from .base import (CodecInterface, VariableComplexityCodecInterface, VariableRateCodecInterface,
                   VariableTaskCodecInterface)


class GroupedVariableRateCodec(HotPathModule, CodecInterface, VariableRateCodecInterface, VariableComplexityCodecInterface,
                               VariableTaskCodecInterface):
    def __init__(self, codecs: List[CodecInterface], *args,
                 codec_vr_level_config: Dict[int, Tuple[int, int]] = None,
                 codec_sc_level_config: Dict[int, Tuple[int, int]] = None, **kwargs):
        super().__init__()
        self.codecs = list(codecs)
        for i, codec in enumerate(self.codecs):
            self.add_module(f"codec_{i}", codec)
        self.codec_vr_level_config = dict(codec_vr_level_config or {})
        self.codec_sc_level_config = dict(codec_sc_level_config or {})   # accepted and stored, unused (as in the reference)
        self.active_codec_idx = 0
        self.set_rate_level(0)

    # ---- selection
    @property
    def active_codec(self):
        return self.codecs[self.active_codec_idx]

    def __len__(self):
        return len(self.codecs)

    def __getitem__(self, index):
        return self.codecs[index]

    def set_rate_level(self, level, *args, **kwargs) -> None:
        member, sub_level = self.codec_vr_level_config.get(level, (level, 0))
        if not 0 <= member < len(self.codecs):
            raise IndexError(f"rate level {level} selects codec {member} of {len(self.codecs)}")
        self.active_codec_idx = member
        if isinstance(self.active_codec, VariableRateCodecInterface):
            self.active_codec.set_rate_level(sub_level, *args, **kwargs)

    @property
    def num_rate_levels(self):
        return len(self.codec_vr_level_config) or len(self.codecs)

    def _targets(self, interface, active_only):
        pool = [self.active_codec] if active_only else self.codecs
        return [c for c in pool if isinstance(c, interface)]

    def set_complex_level(self, level, *args, active_only=False, **kwargs) -> None:
        for codec in self._targets(VariableComplexityCodecInterface, active_only):
            codec.set_complex_level(level, *args, **kwargs)

    def get_current_complex_metrics(self, *args, **kwargs) -> Dict[str, Any]:
        if isinstance(self.active_codec, VariableComplexityCodecInterface):
            return self.active_codec.get_current_complex_metrics(*args, **kwargs)
        return dict()

    @property
    def num_complex_levels(self) -> int:
        return max((c.num_complex_levels if isinstance(c, VariableComplexityCodecInterface) else 0) for c in self.codecs)

    def set_task(self, task, *args, active_only=False, **kwargs) -> bool:
        if active_only:   # (the reference returns None when the active member has no tasks: base.py:189-192)
            if isinstance(self.active_codec, VariableTaskCodecInterface):
                return self.active_codec.set_task(task, *args, **kwargs)
            return None
        success = True
        for codec in self._targets(VariableTaskCodecInterface, False):
            success = bool(success and codec.set_task(task, *args, **kwargs))
        return success

    @property
    def num_tasks(self) -> int:
        return max((c.num_tasks if isinstance(c, VariableTaskCodecInterface) else 0) for c in self.codecs)

    # ---- coding
    def compress(self, data, *args, **kwargs) -> bytes:
        return self.active_codec.compress(data, *args, **kwargs)

    def decompress(self, data: bytes, *args, **kwargs):
        return self.active_codec.decompress(data, *args, **kwargs)

    def forward(self, *args, **kwargs):
        # every member is run (the reference needs that for joint training, base.py:213-221); the active one's result counts
        for codec in self.codecs:
            if codec is not self.active_codec:
                codec(*args, **kwargs)
        return self.active_codec(*args, **kwargs)

    def forward_estimate_bitlen(self, *args, **kwargs):
        return self.active_codec.forward_estimate_bitlen(*args, **kwargs)

    def update_state(self, *args, **kwargs) -> None:
        for codec in self.codecs:
            codec.update_state(*args, **kwargs)

    def post_training_process(self, *args, **kwargs) -> None:
        for codec in self.codecs:
            if hasattr(codec, "post_training_process"):
                codec.post_training_process(*args, **kwargs)

    def load_checkpoint(self, checkpoint_loader=None):   # codecs/base.py:239-243: members first, then the group's own
        for codec in self.codecs:
            if hasattr(codec, "load_checkpoint"):
                codec.load_checkpoint()
        return super().load_checkpoint(checkpoint_loader)
