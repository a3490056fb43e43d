"""Node / parameter generators shared by encoder and decoder -- cbench/nn/layers/param_generator.py:22-70,
213-274,295-328 (NNParameterGenerator, IndexParameterGenerator, IndexSelectParameterGeneratorWrapper), eval path."""
import torch
import torch.nn as nn


class NNParameterGenerator(nn.Module):
    def __init__(self, shape, *args, init_method="zeros", init_value=None, fix_params=False, freeze_params=False, no_params=False, **kwargs):
        super().__init__()
        self.shape, self.no_params = tuple(shape), no_params
        if no_params:
            self.params = None
            return
        init = torch.zeros(self.shape)
        if init_method == "ones":
            init = torch.ones(self.shape)
        elif init_method == "normal":
            init = torch.randn(self.shape)
        elif init_method == "value":
            init = torch.as_tensor(init_value).float().clone()
        elif init_method != "zeros":
            raise NotImplementedError()
        if fix_params:
            self.register_buffer("params", init, persistent=False)
        else:
            self.params = nn.Parameter(init, requires_grad=not freeze_params)

    def forward(self, *args, **kwargs):
        return None if self.no_params else self.params + 0.0


class IndexParameterGenerator(nn.Module):
    def __init__(self, shape, *args, max=1, seed=None, min=0, fix_for_inference=False, fix_for_inference_sample=None, **kwargs):
        super().__init__()
        self.shape, self.max, self.min, self.seed = tuple(shape), max, min, seed
        self.fix_for_inference, self.fix_for_inference_sample = fix_for_inference, fix_for_inference_sample

    @property
    def max_sample(self):
        return self.max - 1

    @property
    def min_sample(self):
        return self.min

    def forward(self, **kwargs):
        if not self.training and self.fix_for_inference:
            return self.min_sample if self.fix_for_inference_sample is None else self.fix_for_inference_sample
        rng = None
        if self.seed is not None:
            rng = torch.Generator()
            rng.manual_seed(self.seed)
        return torch.randint(self.min, self.max, self.shape, generator=rng)


class IndexSelectParameterGeneratorWrapper(IndexParameterGenerator):
    def __init__(self, batched_generator, seed=None, max=None, **kwargs):
        if isinstance(batched_generator, int):
            max = batched_generator
        elif isinstance(batched_generator, (list, tuple)):
            max = len(batched_generator)
            batched_generator = nn.ModuleList(batched_generator)
        elif max is None:
            max = len(batched_generator())
        super().__init__((1,), max=max, seed=seed, **kwargs)
        self.batched_generator = batched_generator

    def forward(self, index=None, **kwargs):
        if index is None:
            index = super().forward(**kwargs)
        if isinstance(index, torch.Tensor):
            index = int(index.reshape(-1)[0].item())
        if isinstance(self.batched_generator, int):
            return index
        if isinstance(self.batched_generator, nn.ModuleList):
            return self.batched_generator[index](**kwargs)
        return self.batched_generator(**kwargs)[index]
