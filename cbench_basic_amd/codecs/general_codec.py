"""GeneralCodec -- cbench/codecs/general_codec.py:18-130,320-346 restricted to the
``entropy_coder``-only configuration every north-star experiment uses
(configs/lossy_graph_scalable_exp_hp.py, configs/lossy_latent_graph_topogroup.py: the latent
graph IS the entropy coder; preprocessor / prior_model / context_model slots are None)."""
from ..base import HotPathModule
from .base import (CodecInterface, VariableComplexityCodecInterface, VariableRateCodecInterface,
                   VariableTaskCodecInterface)


class GeneralCodec(HotPathModule, CodecInterface, VariableRateCodecInterface, VariableComplexityCodecInterface,
                   VariableTaskCodecInterface):
    def __init__(self, *args, preprocessor=None, prior_model=None, context_model=None, entropy_coder=None,
                 prior_first=False, **kwargs):
        super().__init__()
        if preprocessor is not None or prior_model is not None or context_model is not None:
            raise NotImplementedError("only the entropy_coder slot is on the MI355X hot path (SURVEY 8a a1)")
        self.entropy_coder = entropy_coder
        self.prior_first = prior_first

    def compress(self, data, *args, **kwargs):
        # general_codec.py:46-47: the upload is part of compress().  The latent-graph coder does it itself (on its
        # stream, asynchronously from page-locked memory), so a host batch is handed over as it is.
        if data.device != self.device and not (data.device.type == "cpu" and hasattr(self.entropy_coder, "_fused_session")):
            data = data.to(device=self.device)
        with self.profiler.start_time_profile("time_compress_entropy_coder"):
            return self.entropy_coder.encode(data, *args, prior=None, **kwargs)

    def decompress(self, data, *args, **kwargs):
        with self.profiler.start_time_profile("time_decompress_entropy_coder"):
            return self.entropy_coder.decode(data, *args, prior=None, **kwargs)

    def forward(self, data, *args, **kwargs):
        return self.entropy_coder(data, *args, **kwargs)

    def forward_estimate_bitlen(self, data, *args, **kwargs):
        """general_codec.py:190-209: (reconstruction, estimated compressed size in BYTES) from the entropy
        coder's cached rate estimate (prior_entropy is in nats per image)."""
        import math
        import torch
        with torch.no_grad():
            result = self.forward(data, *args, **kwargs)
            estimated_bitlen = 0
            pe = self.entropy_coder.get_raw_cache("metric_dict").get("prior_entropy") \
                if hasattr(self.entropy_coder, "get_raw_cache") else None
            if pe is not None:
                estimated_bitlen = estimated_bitlen + pe * data.size(0) / math.log(2)
            return result, estimated_bitlen / 8

    def update_state(self, *args, **kwargs) -> None:  # general_codec.py:320-326
        for m in self.children():
            if hasattr(m, "update_state"):
                m.update_state(*args, **kwargs)

    def post_training_process(self, *args, **kwargs):
        for m in self.children():
            if hasattr(m, "post_training_process"):
                m.post_training_process(*args, **kwargs)

    # variable rate / complexity / task plumbing (general_codec.py:328-360)
    def set_rate_level(self, level, *args, **kwargs):
        return self.entropy_coder.set_rate_level(level, *args, **kwargs)

    @property
    def num_rate_levels(self):
        return self.entropy_coder.num_rate_levels

    def set_complex_level(self, level, *args, **kwargs):
        return self.entropy_coder.set_complex_level(level, *args, **kwargs)

    def get_current_complex_metrics(self, *args, **kwargs):
        return self.entropy_coder.get_current_complex_metrics(*args, **kwargs)

    @property
    def num_complex_levels(self):
        return self.entropy_coder.num_complex_levels

    def set_task(self, task, *args, **kwargs):
        return self.entropy_coder.set_task(task, *args, **kwargs)

    @property
    def num_tasks(self):
        return self.entropy_coder.num_tasks
