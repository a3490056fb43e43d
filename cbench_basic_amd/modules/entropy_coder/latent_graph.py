"""Latent-graph entropy model driver -- API of LatentGraphicalANSEntropyCoder
(cbench/modules/entropy_coder/latent_graph.py:306) restricted to the encode / decode /
update_state hot path:

    encode:  node generators -> inference edges (x->y->z: g_a, h_a) -> generative pass
             (z then y): per node  forward (quantise)  +  encode (bytes),  generative edges
             (z->y: h_s)                                     latent_graph.py:1232-1264, :721, :760
    decode:  split bytes -> generative pass z -> y -> x (h_s, g_s)      latent_graph.py:1266-1295
    stream:  merge_bytes([bytes_z, bytes_y], num_segments=2)            utils/bytes_ops.py:19-33

Graph traversal is tiny host logic; every tensor operation it triggers is a HIP kernel.
Training-only machinery (losses, MC sampling, FLOPs regularisers, sandwich rule) is out of
scope.  The greedy complexity-level search of BaSIC (post_training_process, latent_graph.py:1397-1640) is here:
selection logic in complexity_search.py, evaluation = this codec's own forward on the GPU; a ready-made result
(per-level node parameters) can also be installed with ``set_complexity_level_params``.
"""
from typing import Any, Dict, Iterable, List, Optional

import math
import time
import torch
import torch.nn as nn

from ...base import HotPathModule
from ...codecs.base import (VariableComplexityCodecInterface, VariableRateCodecInterface,
                            VariableTaskCodecInterface)
from ...utils.bytes_ops import merge_bodies, merge_bytes, split_merged_bytes, split_merged_views
from .complexity_search import search_complexity_levels


class LossyDummyEntropyCoder(HotPathModule):
    """latent_graph.py:68-144: the input node of a lossy codec carries no bits; decoding
    returns the prior (= g_s output)."""

    def __init__(self, *args, lambda_rd=1.0, distortion_type="mse", **kwargs):
        super().__init__()
        self.lambda_rd = lambda_rd
        self.distortion_type = distortion_type

    def forward(self, data, *args, prior=None, lambda_rd=None, prior_target=None, **kwargs):
        """latent_graph.py:121-141: distortion metrics of the reconstruction (= prior).  metric_dict gets ``mse`` and
        ``weighted_distortion`` = lambda_rd * (sum of squared errors per image, averaged over the batch) (:83-88,:139)."""
        if isinstance(data, torch.Tensor) and isinstance(prior, torch.Tensor):
            if self.distortion_type not in ("mse", "ms-ssim"):
                raise NotImplementedError(f"distortion_type {self.distortion_type}")
            target = data if prior_target is None else prior_target
            rec = prior
            for dim, size in enumerate(target.shape[2:], 2):  # crop to the target size (:127-129)
                if rec.shape[dim] != size:
                    rec = rec.narrow(dim, 0, size)
            if self.distortion_type == "ms-ssim":
                # :92-96 (the "...-ft-ssim" presets): loss = mean over the batch of (1 - MS-SSIM) x elements per image; the metric
                # comes from benchmark/ms_ssim.py, a restatement of the absent pytorch_msssim package (parity-unpinned)
                from ...benchmark.ms_ssim import ms_ssim
                val = ms_ssim(rec, target, data_range=1.0, size_average=False)
                loss_distortion = (1 - val).mean() * (target.numel() // target.shape[0])
                self.update_cache("metric_dict", ms_ssim=val.mean(),
                                  weighted_distortion=loss_distortion * (self.lambda_rd if lambda_rd is None else lambda_rd))
                return rec
            from ...nn import kernels as K
            mse = K.mse_per_image(rec.contiguous(), target.contiguous())  # [B], HIP reduction
            loss_distortion = (mse * (target.numel() // target.shape[0])).mean()
            self.update_cache("metric_dict", mse=mse.mean(),
                              weighted_distortion=loss_distortion * (self.lambda_rd if lambda_rd is None else lambda_rd))
            return rec     # the reference returns the NARROWED prior (:121-123,:141): forward() is input-sized, decode() is not
        return prior

    def encode(self, data, *args, prior=None, **kwargs) -> bytes:
        return b""

    def decode(self, byte_string: bytes, *args, prior=None, **kwargs):
        return prior

    def update_state(self, *args, **kwargs):
        pass


class ParamDictModuleWrapper(nn.Module):
    """latent_graph.py:270-298: keeps a dict of node values as buffers so the searched complexity levels travel in the
    state_dict under the reference's keys (``_complexity_param_all_levels.<level>.<node>``)."""

    def __init__(self, params: Dict[str, Any]):
        super().__init__()
        self.none_params = []
        for name, value in params.items():
            if value is None:
                self.none_params.append(name)
            elif isinstance(value, dict):
                setattr(self, name, ParamDictModuleWrapper(value))
            else:
                self.register_buffer(name, torch.as_tensor(value).detach().clone())

    def forward(self, *args, **kwargs):
        out = {name: None for name in self.none_params}
        out.update(self._buffers)
        for name, m in self._modules.items():
            out[name] = m()
        return out


class BasicLatentGraphicalNodeAggregatorModel(nn.Module):
    """latent_graph.py:55-60: folds the list of values several edges delivered to one node."""

    def forward(self, input_list, *args, **kwargs):
        raise NotImplementedError()


class AverageNodeAggregatorModel(BasicLatentGraphicalNodeAggregatorModel):
    """latent_graph.py:63-65."""

    def forward(self, input_list, *args, **kwargs):
        return torch.stack(input_list).mean(0)


class LatentGraphicalANSEntropyCoder(HotPathModule, VariableRateCodecInterface, VariableComplexityCodecInterface,
                                     VariableTaskCodecInterface):
    DEFAULT_EDGE_SPLIT_SYMBOL = "_"
    DEFAULT_PRIOR_KEY_NAME = "prior"
    DEFAULT_UNCONDITIONAL_NODE_NAME = "u"
    DEFAULT_INPUT_NODE_NAME = "x"
    DEFAULT_RATE_LEVEL_NODE_NAME = "vrlevel"
    DEFAULT_COMPLEX_LEVEL_NODE_NAME = "sclevel"
    DEFAULT_TASK_INDEX_NODE_NAME = "taskidx"

    def __init__(self, *args,
                 use_lossy_compression=True,
                 lossy_compression_lambda_rd=1.0,
                 lossy_compression_distortion_type="mse",
                 node_generator_dict: Dict[str, nn.Module] = None,
                 node_generator_input_mapping: Optional[Dict[str, Dict[str, str]]] = None,
                 dynamic_node_generator_dict: Dict[str, nn.Module] = None,
                 latent_node_entropy_coder_dict: Dict[str, nn.Module] = None,
                 latent_inference_dict: Dict[str, nn.Module] = None,
                 latent_generative_dict: Dict[str, nn.Module] = None,
                 latent_inference_input_mapping: Optional[Dict[str, Dict[str, str]]] = None,
                 latent_generative_input_mapping: Optional[Dict[str, Dict[str, str]]] = None,
                 latent_node_inference_topo_order: Optional[List[str]] = None,
                 latent_node_generative_topo_order: Optional[List[str]] = None,
                 latent_inference_node_aggregator_dict: Optional[Dict[str, nn.Module]] = None,
                 latent_generative_node_aggregator_dict: Optional[Dict[str, nn.Module]] = None,
                 complexity_metric_list=None,
                 complexity_level_greedy_search=False,
                 complexity_level_greedy_search_dataset: Optional[Iterable] = None,
                 complexity_level_greedy_search_dataset_cached=False,
                 complexity_level_greedy_search_iterative=False,
                 complexity_level_greedy_search_num_levels: Optional[int] = None,
                 complexity_level_greedy_search_custom_constraint: Optional[List[float]] = None,
                 complexity_level_greedy_search_performance_metric: Optional[str] = None,
                 complexity_level_greedy_search_complexity_metric: Optional[str] = None,
                 complexity_level_greedy_search_add_controller_nodes_as_complexity_metric=True,
                 complexity_level_controller_nodes=(),
                 complexity_level_greedy_search_custom_params: Optional[List[Dict[str, int]]] = None,
                 complexity_level_greedy_search_loss_mode="eval",
                 task_names: Optional[List[str]] = None,
                 **kwargs):
        super().__init__()
        node_generator_dict = dict(node_generator_dict or {})
        dynamic_node_generator_dict = dict(dynamic_node_generator_dict or {})
        latent_node_entropy_coder_dict = dict(latent_node_entropy_coder_dict or {})
        self.use_lossy_compression = use_lossy_compression
        if use_lossy_compression and self.DEFAULT_INPUT_NODE_NAME not in latent_node_entropy_coder_dict:
            latent_node_entropy_coder_dict[self.DEFAULT_INPUT_NODE_NAME] = LossyDummyEntropyCoder(
                lambda_rd=lossy_compression_lambda_rd, distortion_type=lossy_compression_distortion_type)

        self.node_generators = nn.ModuleDict(node_generator_dict)
        self.node_generator_input_mapping = dict(node_generator_input_mapping or {})
        self.dynamic_node_generators = nn.ModuleDict(dynamic_node_generator_dict)
        self.latent_node_entropy_coders = nn.ModuleDict(latent_node_entropy_coder_dict)
        self.latent_inference_modules = nn.ModuleDict(dict(latent_inference_dict or {}))
        self.latent_generative_modules = nn.ModuleDict(dict(latent_generative_dict or {}))
        self.latent_inference_input_mapping = dict(latent_inference_input_mapping or {})
        self.latent_generative_input_mapping = dict(latent_generative_input_mapping or {})
        # multi-edge aggregators (latent_graph.py:343-344,467-468): a node fed by SEVERAL edges collects their outputs in a list,
        # the node's aggregator folds it.  Same attribute names as the reference: they are state_dict prefixes
        self.latent_inference_node_aggregator_modules = nn.ModuleDict(dict(latent_inference_node_aggregator_dict or {}))
        self.latent_generative_node_aggregator_modules = nn.ModuleDict(dict(latent_generative_node_aggregator_dict or {}))
        self.latent_node_inference_topo_order = list(latent_node_inference_topo_order)
        self.latent_node_generative_topo_order = list(latent_node_generative_topo_order)
        self.complexity_level_greedy_search = complexity_level_greedy_search
        self.complexity_level_controller_nodes = list(complexity_level_controller_nodes)
        self.task_names = task_names

        # edges grouped by destination / source node (latent_graph.py:546-558)
        sym = self.DEFAULT_EDGE_SPLIT_SYMBOL
        self._inference_edges_into = {n: [] for n in self.latent_node_inference_topo_order}
        for edge in self.latent_inference_modules:
            self._inference_edges_into[edge.split(sym)[1]].append(edge)
        self._generative_edges_from = {n: [] for n in self.latent_node_generative_topo_order}
        for edge in self.latent_generative_modules:
            self._generative_edges_from[edge.split(sym)[0]].append(edge)

        # variable rate / complexity / task bookkeeping (latent_graph.py:566-648)
        def _levels(name):
            if name in self.dynamic_node_generators:
                g = self.dynamic_node_generators[name]
                return g.max_sample - g.min_sample + 1
            return 0
        self._num_rate_levels = _levels(self.DEFAULT_RATE_LEVEL_NODE_NAME)
        self._num_tasks = _levels(self.DEFAULT_TASK_INDEX_NODE_NAME)
        self._num_complex_levels = _levels(self.DEFAULT_COMPLEX_LEVEL_NODE_NAME)
        self._current_rate_level = -1
        self._current_task_idx = -1
        self._current_complex_level = -1
        self._complexity_param_all_levels = None  # ModuleList of ParamDictModuleWrapper, one per level
        self._searching = False  # True while post_training_process evaluates explicit controller settings
        self._valid_host = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: setattr(module, "_valid_host", None))
        # greedy complexity-level search (latent_graph.py:590-634)
        self.complexity_metric_list = list(complexity_metric_list or [])
        self.complexity_level_greedy_search_dataset = complexity_level_greedy_search_dataset
        self.complexity_level_greedy_search_dataset_cached = complexity_level_greedy_search_dataset_cached
        self.complexity_level_greedy_search_iterative = complexity_level_greedy_search_iterative
        self.complexity_level_greedy_search_custom_constraint = complexity_level_greedy_search_custom_constraint
        self.complexity_level_greedy_search_custom_params = complexity_level_greedy_search_custom_params
        # "eval": the loss on the rounded latents the codec codes (deterministic, the default here); "train": the reference's
        # own evaluation (latent_graph.py:1322 self.train()): additive-uniform-noise proxies in every coder, a random variable
        if complexity_level_greedy_search_loss_mode not in ("eval", "train"):
            raise ValueError("complexity_level_greedy_search_loss_mode: 'eval' or 'train'")
        self.complexity_level_greedy_search_loss_mode = complexity_level_greedy_search_loss_mode
        self.complexity_level_greedy_search_complexity_metric = complexity_level_greedy_search_complexity_metric
        self.complexity_level_greedy_search_performance_metric = complexity_level_greedy_search_performance_metric
        if complexity_level_greedy_search:
            if complexity_level_greedy_search_num_levels is not None:
                self._num_complex_levels = complexity_level_greedy_search_num_levels
            elif complexity_level_greedy_search_custom_constraint is not None:
                self._num_complex_levels = len(complexity_level_greedy_search_custom_constraint)
            valid = False
            if complexity_level_greedy_search_custom_params is not None:
                levels = [{k: self.node_generators[k](v) for k, v in idx.items()}
                          for idx in complexity_level_greedy_search_custom_params]
                self._num_complex_levels = len(levels)
            else:  # every level starts as the most complex setting until post_training_process has run (:604-615)
                most = {k: self.node_generators[k](self.node_generators[k].min_sample) for k in self.complexity_level_controller_nodes}
                levels = [most for _ in range(self._num_complex_levels)]
            self._complexity_param_all_levels = nn.ModuleList([ParamDictModuleWrapper(lv) for lv in levels])
            self.register_buffer("_complexity_param_valid", torch.tensor([valid]))
            if self.complexity_level_greedy_search_complexity_metric is None:
                self.complexity_level_greedy_search_complexity_metric = "FLOPs"
            self.complexity_metric_list.append(self.complexity_level_greedy_search_complexity_metric)
            if self.complexity_level_greedy_search_performance_metric is None:
                self.complexity_level_greedy_search_performance_metric = "loss"
            self.complexity_metric_list.append(self.complexity_level_greedy_search_performance_metric)
            if complexity_level_greedy_search_add_controller_nodes_as_complexity_metric:
                self.complexity_metric_list.extend(self.complexity_level_controller_nodes)
        if self._num_complex_levels > 0 and len(self.complexity_metric_list) > 0:  # (:641-645)
            self.register_buffer("_complexity_metric_list_cache",
                                 torch.zeros(self._num_complex_levels, len(self.complexity_metric_list)))
        if self._num_tasks > 0 and self.task_names is None:
            self.task_names = list(range(self._num_tasks))

    # Optional callable run by encode() right after the analysis transforms have been ENQUEUED on the current HIP stream
    # (before the entropy stage): concurrent stream workers use it to order their transform phases (benchmark/stream_workers.py).
    after_inference_hook = None

    def _levels_valid(self):
        """_complexity_param_valid as a host bool (read once; refreshed after load_state_dict / search)."""
        if self._valid_host is None:
            self._valid_host = bool(self._complexity_param_valid.item())
        return self._valid_host

    # ---- default nodes (latent_graph.py:650-683)
    def _get_default_node_dict(self, force_add_default_dynamic_nodes=False, **kwargs):
        out = {self.DEFAULT_UNCONDITIONAL_NODE_NAME: None, **kwargs}
        if self.training:
            return out
        if self._num_rate_levels > 0 and self.DEFAULT_RATE_LEVEL_NODE_NAME not in out:
            if self._current_rate_level >= 0:
                out[self.DEFAULT_RATE_LEVEL_NODE_NAME] = self._current_rate_level
            elif force_add_default_dynamic_nodes:
                out[self.DEFAULT_RATE_LEVEL_NODE_NAME] = 0
        if self._num_tasks > 0 and self.DEFAULT_TASK_INDEX_NODE_NAME not in out:
            if self._current_task_idx >= 0:
                out[self.DEFAULT_TASK_INDEX_NODE_NAME] = self._current_task_idx
            elif force_add_default_dynamic_nodes:
                out[self.DEFAULT_TASK_INDEX_NODE_NAME] = 0
        if self._num_complex_levels > 0 and self.DEFAULT_COMPLEX_LEVEL_NODE_NAME not in out:
            if self.complexity_level_greedy_search and self._levels_valid() and not self._searching:
                out.update(**self._complexity_param_all_levels[self._current_complex_level]())  # :673-675
            elif self._current_complex_level >= 0:
                out[self.DEFAULT_COMPLEX_LEVEL_NODE_NAME] = self._current_complex_level
            elif force_add_default_dynamic_nodes:
                out[self.DEFAULT_COMPLEX_LEVEL_NODE_NAME] = 0
        return out

    # ---- shared encoder/decoder constants (latent_graph.py:686-719)
    def _node_generate_process(self, **kwargs):
        out = dict(**kwargs)
        sym = self.DEFAULT_EDGE_SPLIT_SYMBOL
        for name, module in self.node_generators.items():
            inputs = {ik: out[nk] for nk, ik in self.node_generator_input_mapping.get(name, {}).items()}
            if sym in name:
                inode, onode = name.split(sym)
                if onode in out:
                    continue
                if name in out:
                    out[onode] = out.pop(name)
                    continue
                out[onode] = module(out[inode], **inputs)
            elif name not in out:
                out[name] = module(**inputs)
        return out

    # ---- inference pass (latent_graph.py:721-758)
    def _inference_process(self, input_dict):
        out = dict(**input_dict)
        sym = self.DEFAULT_EDGE_SPLIT_SYMBOL
        for node in self.latent_node_inference_topo_order:
            kw = {ik: out[nk] for nk, ik in self.latent_inference_input_mapping.get(node, {}).items()}
            for edge in self._inference_edges_into[node]:
                kw.update({ik: out[ek] for ek, ik in self.latent_inference_input_mapping.get(edge, {}).items()})
                inode, onode = edge.split(sym)
                with self.profiler.start_time_profile(f"latent_inference_modules_{edge}"):
                    val = self.latent_inference_modules[edge](out[inode], **kw)
                if onode in out:   # a second edge into the node: collect (latent_graph.py:741-746)
                    if not isinstance(out[onode], list):
                        out[onode] = [out[onode]]
                    out[onode].append(val)
                else:
                    out[onode] = val
            if node in self.latent_inference_node_aggregator_modules:   # (:748-749)
                out[node] = self.latent_inference_node_aggregator_modules[node](out[node])
        return out

    # ---- generative pass (latent_graph.py:760-868)
    def _generative_process(self, input_dict, prior_dict=None, do_encode=False, **kwargs):
        data = dict(**input_dict)
        kw_all = dict(**input_dict, **kwargs)
        prior_dict = dict(prior_dict or {})
        sym = self.DEFAULT_EDGE_SPLIT_SYMBOL
        nodes = list(self.latent_node_generative_topo_order)
        if do_encode and self.use_lossy_compression:
            nodes.remove(self.DEFAULT_INPUT_NODE_NAME)
        for node in nodes:
            prior_dict.setdefault(node, dict())
            if node in self.latent_generative_node_aggregator_modules:
                # (:795-796) the reference replaces the node's prior DICT by the aggregator's result and then goes on treating it
                # as a dict (len(), .values(), :806-819): only an aggregator that returns a mapping works there; same here
                agg = self.latent_generative_node_aggregator_modules[node](list(prior_dict[node].values()))
                if not isinstance(agg, dict):
                    raise TypeError(f"latent_generative_node_aggregator of node {node} must return a dict of priors "
                                    "(the reference's own traversal reads .values() of the result, latent_graph.py:806-819)")
                prior_dict[node] = agg
            node_data = data.get(node)
            if node in self.latent_node_entropy_coders:
                coder = self.latent_node_entropy_coders[node]
                pk = dict()
                priors = prior_dict[node]
                if node in self.latent_generative_input_mapping:
                    for nk, ik in self.latent_generative_input_mapping[node].items():
                        if len(priors) == 1:
                            pk.update(prior=list(priors.values())[0])
                        if nk in priors:
                            pk[ik] = priors[nk]
                        elif nk in kw_all:
                            pk[ik] = kw_all[nk]
                        else:
                            raise ValueError(f"latent_generative_input_mapping incorrect! node {node}, mapping {nk} : {ik}")
                else:
                    assert len(priors) <= 1
                    if len(priors) == 1:
                        pk.update(prior=list(priors.values())[0])
                with self.profiler.start_time_profile(f"latent_node_entropy_coders_{node}"):
                    if isinstance(node_data, (bytes, memoryview)):
                        node_data = coder.decode(node_data, **pk)
                        data[node] = node_data
                    else:
                        raw = node_data
                        # forward: quantised latent (:836).  When encoding, a node whose quantised value feeds no
                        # further module (the last latent: its only edge is the synthesis transform, skipped
                        # below) does not need it, and nobody reads the rate estimate either.
                        feeds = [e for e in self._generative_edges_from[node]
                                 if not (do_encode and self.use_lossy_compression and e.split(sym)[1] == self.DEFAULT_INPUT_NODE_NAME)]
                        if do_encode and not feeds:
                            node_data = None
                        else:
                            node_data = coder(raw, **pk)
                        if do_encode:  # (:838); coders that can, only enqueue their GPU work here (resolved in encode())
                            data[node] = coder.encode(raw, lazy=True, **pk) if getattr(coder, "supports_lazy_encode", False) \
                                else coder.encode(raw, **pk)
                        else:
                            data[node] = node_data
                    kw_all[node] = node_data
            for edge in self._generative_edges_from[node]:
                ek = {ik: kw_all[k] for k, ik in self.latent_generative_input_mapping.get(edge, {}).items()}
                inode, onode = edge.split(sym)
                if do_encode and self.use_lossy_compression and onode == self.DEFAULT_INPUT_NODE_NAME:
                    continue  # g_s is not needed to encode (:855-856)
                with self.profiler.start_time_profile(f"latent_generative_modules_{edge}"):
                    val = self.latent_generative_modules[edge](node_data, **ek)
                kw_all[edge] = val
                prior_dict.setdefault(onode, dict())[inode] = val
        return data, prior_dict

    def _coded_nodes(self):
        nodes = list(self.latent_node_generative_topo_order)
        if self.use_lossy_compression:
            nodes.remove(self.DEFAULT_INPUT_NODE_NAME)
        return nodes

    # ---- fused C entry points for the plain hyperprior graph (include/basic_hip.h section 8)
    use_fused_session = True    # set False to force the module-by-module path (the two give identical bytes: tests)
    fused_rans_waves = 0        # wavefronts (image streams) per workgroup of the session's rANS launches; 0 = library default
    fused_transform_token = False   # order the transform phases of all such sessions in GPU time (stream workers): True / 1 = one at a time, 2 = two at a time

    def _fused_session(self, kwargs, prior):
        """The HyperpriorSession serving this graph, or None when the graph is anything but
        x -g_a-> y -h_a-> z, z: EntropyBottleneck coder, y: GaussianConditional coder on h_s(z), x: dummy
        (configs/lossy_graph_scalable_exp_hp.py:182-215) with no dynamic nodes / gains / caller-supplied inputs."""
        if not self.use_fused_session or self.training or prior is not None or kwargs:
            return None
        elig = getattr(self, "_fused_eligible", None)
        if elig is None:
            from ..prior_model.prior_coder.compressai_coder import (CompressAIEntropyBottleneckPriorCoder,
                                                                    CompressAIGaussianConditionalCoder)
            from ...nn.models.google import BasicHyperpriorModule
            c, inf, gen = self.latent_node_entropy_coders, self.latent_inference_modules, self.latent_generative_modules
            elig = (list(self.latent_node_inference_topo_order) == ["x", "y", "z"]
                    and list(self.latent_node_generative_topo_order) == ["z", "y", "x"]
                    and len(self.node_generators) == 0 and self._num_rate_levels <= 0 and self._num_complex_levels <= 0
                    and self._num_tasks <= 0 and set(inf.keys()) == {"x_y", "y_z"} and set(gen.keys()) == {"z_y", "y_x"}
                    and all(type(m).forward is BasicHyperpriorModule.forward for m in list(inf.values()) + list(gen.values()))
                    and type(c["y"]) is CompressAIGaussianConditionalCoder and type(c["z"]) is CompressAIEntropyBottleneckPriorCoder
                    and not self.latent_inference_input_mapping and not self.latent_generative_input_mapping
                    and (self.use_lossy_compression or isinstance(c["x"] if "x" in c else None, LossyDummyEntropyCoder)))
            self._fused_eligible = bool(elig)
        if not elig:
            return None
        c, inf, gen = self.latent_node_entropy_coders, self.latent_inference_modules, self.latent_generative_modules
        zc, yc = c["z"], c["y"]
        zc._ready()
        yc._ready()
        plans = [m.plans() for m in (inf["x_y"], inf["y_z"], gen["z_y"], gen["y_x"])]
        key = tuple(p._h.value for pl in plans for p in pl) + (zc._tables._h.value, yc._tables._h.value)
        sess = getattr(self, "_fused", None)
        if sess is None or sess.key != key:
            from ...nn import kernels as K
            sess = K.HyperpriorSession(*plans, zc.entropy_bottleneck.medians(), zc._tables, yc.scale_table, yc.scale_bound, yc._tables)
            self._fused = sess
        if getattr(sess, "_waves", 0) != self.fused_rans_waves:
            sess.set_rans_waves(self.fused_rans_waves)
            sess._waves = self.fused_rans_waves
        if getattr(sess, "_token", False) != self.fused_transform_token:
            sess.set_transform_token(self.fused_transform_token)
            sess._token = self.fused_transform_token
        return sess

    def encode(self, data, *args, prior=None, **kwargs):
        with torch.no_grad():
            sess = self._fused_session(kwargs, prior) if not args else None
            if sess is not None:   # one C call: upload (when the batch is on the host), transforms, entropy stage, framing
                if data.is_cuda and data.device != self.device:
                    data = data.to(device=self.device)
                with self.profiler.start_time_profile("encode_fused"):
                    out = sess.encode(data)
                if self.after_inference_hook is not None:
                    self.after_inference_hook()
                return out
            if data.device != self.device:
                data = data.to(device=self.device)
            node_dict = self._node_generate_process(**self._get_default_node_dict(force_add_default_dynamic_nodes=True, **kwargs))
            input_dict = {self.DEFAULT_INPUT_NODE_NAME: data, **node_dict}
            prior_dict = dict() if prior is None else {self.DEFAULT_INPUT_NODE_NAME: dict(prior=prior)}
            with self.profiler.start_time_profile("encode_inference"):
                latent_dict = self._inference_process(input_dict)
            if self.after_inference_hook is not None:   # the analysis transforms are enqueued (see stream_workers.py)
                self.after_inference_hook()
            with self.profiler.start_time_profile("encode_generative"):
                data_dict, _ = self._generative_process(latent_dict, prior_dict=prior_dict, do_encode=True)
            nodes = self._coded_nodes()
            bodies = [data_dict[n] for n in nodes]
            if any(hasattr(b, "write_into") for b in bodies):
                return merge_bodies(bodies)  # = merge_bytes(..., num_segments=len(nodes)), bodies framed in place
            bodies = [b.result() if hasattr(b, "result") else b for b in bodies]
            return merge_bytes(bodies, num_segments=len(nodes))

    def decode(self, data, *args, prior=None, **kwargs):
        with torch.no_grad():
            sess = self._fused_session(kwargs, prior) if not args else None
            if sess is not None:
                with self.profiler.start_time_profile("decode_fused"):
                    return sess.decode(data, device=self.device)
            node_dict = self._node_generate_process(**self._get_default_node_dict(force_add_default_dynamic_nodes=True, **kwargs))
            nodes = self._coded_nodes()
            # coders that read their stream through the buffer protocol get it in place; others get bytes
            segs = split_merged_views(data, num_segments=len(nodes))
            input_dict = {n: (v if getattr(self.latent_node_entropy_coders[n], "accepts_buffer", False) else v.tobytes())
                          for n, v in zip(nodes, segs)}
            prior_dict = dict() if prior is None else {self.DEFAULT_INPUT_NODE_NAME: dict(prior=prior)}
            if self.use_lossy_compression:
                input_dict[self.DEFAULT_INPUT_NODE_NAME] = b""
            with self.profiler.start_time_profile("decode_generative"):
                data_dict, _ = self._generative_process(input_dict, prior_dict=prior_dict, **node_dict)
            return data_dict[self.DEFAULT_INPUT_NODE_NAME]

    def forward(self, data, *args, **kwargs):
        """Eval-mode forward: quantised reconstruction without entropy coding (latent_graph.py:870-1230
        minus the training losses)."""
        with torch.no_grad():
            node_dict = self._node_generate_process(**self._get_default_node_dict(force_add_default_dynamic_nodes=True, **kwargs))
            latent = self._inference_process({self.DEFAULT_INPUT_NODE_NAME: data.to(self.device), **node_dict})
            # the coders' forward() inside encode() skips the likelihood pass (nobody reads it there); here it is the
            # point of the call (prior_entropy / estimated_bpd), as in the reference's forward
            coders = [c for c in self.latent_node_entropy_coders.values() if hasattr(c, "estimate_rate")]
            saved = [c.estimate_rate for c in coders]
            for c in coders:
                c.estimate_rate = True
            try:
                data_dict, prior_dict = self._generative_process(latent)
            finally:
                for c, v in zip(coders, saved):
                    c.estimate_rate = v
            # rate metrics (latent_graph.py:1168-1178): nats per image summed over the coded nodes, bits per dimension
            total_prior_entropy, estimated_bpd = 0, 0
            for name, module in self.latent_node_entropy_coders.items():
                pe = module.get_raw_cache("metric_dict").get("prior_entropy") if hasattr(module, "get_raw_cache") else None
                if pe is not None:
                    total_prior_entropy = total_prior_entropy + pe
                    estimated_bpd = estimated_bpd + pe / math.log(2) / (data.numel() / data.size(0))
            self.update_cache("metric_dict", prior_entropy=total_prior_entropy, estimated_bpd=estimated_bpd)
            return data_dict[self.DEFAULT_INPUT_NODE_NAME]

    def update_state(self, *args, **kwargs) -> None:  # latent_graph.py:1297-1301
        for coder in self.latent_node_entropy_coders.values():
            coder.update_state(*args, **kwargs)

    # ---- VariableRate / Complexity / Task (latent_graph.py:1660-1691)
    def set_rate_level(self, level, *args, **kwargs) -> None:
        assert 0 <= level < max(1, self._num_rate_levels)
        self._current_rate_level = level

    @property
    def num_rate_levels(self):
        return max(1, self._num_rate_levels)

    def set_complex_level(self, level, *args, **kwargs) -> None:
        assert 0 <= level < max(1, self._num_complex_levels)
        self._current_complex_level = level

    @property
    def num_complex_levels(self):
        return max(1, self._num_complex_levels)

    def set_complexity_level_params(self, params_per_level: List[Dict[str, Any]]):
        """Install a ready-made outcome of the search (latent_graph.py:1615-1619): one dict of controller-node values
        (e.g. slim one-hots pgmxy/pgmyz/pgmzy/pgmyx) per level."""
        dev = self.device
        self._complexity_param_all_levels = nn.ModuleList([ParamDictModuleWrapper(p) for p in params_per_level]).to(dev)
        self._num_complex_levels = len(params_per_level)
        self.complexity_level_greedy_search = True
        if not hasattr(self, "_complexity_param_valid"):
            self.register_buffer("_complexity_param_valid", torch.tensor([True], device=dev))
        self._complexity_param_valid.fill_(True)
        self._valid_host = None
        if len(self.complexity_metric_list) > 0:
            self.register_buffer("_complexity_metric_list_cache",
                                 torch.zeros(self._num_complex_levels, len(self.complexity_metric_list), device=dev))

    def get_current_complex_metrics(self, *args, **kwargs):  # latent_graph.py:1675-1687
        if not hasattr(self, "_complexity_metric_list_cache"):
            return dict()
        row = self._complexity_metric_list_cache[self._current_complex_level]
        return {name: row[i].item() for i, name in enumerate(self.complexity_metric_list)}

    # ---- complexity accounting and the greedy level search (latent_graph.py:1303-1640)
    def get_current_flops(self, input=None):
        """Sum of the dynamic transforms' operation counters of the last forward (nn/base.py:675-680)."""
        total = 0.0
        for m in self.modules():
            if m is not self and hasattr(m, "get_current_flops"):
                total += float(m.get_current_flops())
        return total

    def _test_dataset_complexity_performance(self, dataset, *args, performance_method="loss", complexity_method="FLOPs",
                                             loss_mode=None, **node_params):
        """(complexity, performance) of one controller setting over a dataset, both divided by the number of input
        elements (latent_graph.py:1320-1395).

        performance "loss" = the rate and distortion loss terms: loss_rate = prior_entropy / ln 2 of every coded node
        (bits per image; compressai_coder.py:213-215, pgm_coder.py:516) + loss_distortion = lambda_rd * SSE per image
        (:83-88).  The reference evaluates them in train mode, i.e. with its additive-uniform-noise proxies, so its
        number is a random variable (loss_mode "train": every coder's forward adds U(-.5, .5) instead of rounding --
        torch_ans.py:127-136 and upstream quantize(..., "noise") -- drawn from torch's CUDA generator); the default
        (loss_mode "eval", or complexity_level_greedy_search_loss_mode) evaluates them on the ROUNDED latents the codec
        actually codes, which is deterministic.  complexity "FLOPs" = get_current_flops(); the timing variants
        ("compress_time", "decompress_time", "total_time") wall-clock encode / decode."""
        if performance_method != "loss":
            raise NotImplementedError(performance_method)
        coders = [c for c in self.latent_node_entropy_coders.values() if hasattr(c, "estimate_rate")]
        saved = [c.estimate_rate for c in coders]
        for c in coders:
            c.estimate_rate = True
        loss_mode = loss_mode or self.complexity_level_greedy_search_loss_mode
        proxied = [m for c in self.latent_node_entropy_coders.values() if isinstance(c, nn.Module) for m in c.modules()]
        for m in proxied:
            m.rate_proxy = "noise" if loss_mode == "train" else "round"
        performance, complexity, total_dims = 0.0, 0.0, 0
        searching, self._searching = self._searching, True
        try:
            for data in dataset:
                if isinstance(data, (list, tuple)):
                    data = data[0]
                total_dims += data.numel()
                self.forward(data, *args, **node_params)
                loss = 0.0
                for name, coder in self.latent_node_entropy_coders.items():
                    md = coder.get_raw_cache("metric_dict") if hasattr(coder, "get_raw_cache") else {}
                    if "prior_entropy" in md:
                        loss += float(md["prior_entropy"]) / math.log(2)
                    if "weighted_distortion" in md:
                        loss += float(md["weighted_distortion"])
                performance += loss
                if complexity_method == "FLOPs":
                    complexity += self.get_current_flops()
                elif complexity_method in ("compress_time", "decompress_time", "total_time"):
                    torch.cuda.synchronize()
                    t0 = time.time()
                    byte_string = self.encode(data, *args, **node_params)
                    t1 = time.time()
                    out = self.decode(byte_string, *args, **node_params)
                    if isinstance(out, torch.Tensor) and out.is_cuda:
                        torch.cuda.synchronize()
                    t2 = time.time()
                    complexity += {"compress_time": t1 - t0, "decompress_time": t2 - t1, "total_time": t2 - t0}[complexity_method]
                else:
                    raise NotImplementedError(complexity_method)
                self.reset_all_cache()
        finally:
            self._searching = searching
            for m in proxied:
                m.rate_proxy = "round"
            for c, v in zip(coders, saved):
                c.estimate_rate = v
        if total_dims == 0:
            raise ValueError("complexity_level_greedy_search_dataset is empty")
        return complexity / total_dims, performance / total_dims

    def post_training_process(self, *args, force=False, dataset=None, **kwargs) -> None:
        """latent_graph.py:1397-1640: fill the complexity levels by searching the controller settings.  Runs once
        (``_complexity_param_valid``) unless ``force``; ``dataset`` overrides complexity_level_greedy_search_dataset."""
        if not self.complexity_level_greedy_search:
            return
        self.update_state()
        if not (force or not self._levels_valid()):
            return
        names = list(self.complexity_level_controller_nodes)
        if self.complexity_level_greedy_search_custom_params is not None:
            levels = [dict(idx) for idx in self.complexity_level_greedy_search_custom_params]
            rows = None
        else:
            if self.complexity_level_greedy_search_iterative:
                raise NotImplementedError("complexity_level_greedy_search_iterative (see complexity_search.py)")
            data = dataset if dataset is not None else self.complexity_level_greedy_search_dataset
            if data is None:
                raise ValueError("complexity_level_greedy_search_dataset is required for the complexity search")
            if self.complexity_level_greedy_search_dataset_cached or not isinstance(data, (list, tuple)):
                data = list(data)
            gens = {n: self.node_generators[n] for n in names}

            def evaluate(idx):
                return self._test_dataset_complexity_performance(
                    data, performance_method=self.complexity_level_greedy_search_performance_metric,
                    complexity_method=self.complexity_level_greedy_search_complexity_metric,
                    **{n: gens[n](idx[n]) for n in names})

            result = search_complexity_levels(
                evaluate, names, {n: gens[n].min_sample for n in names}, {n: gens[n].max_sample for n in names},
                num_levels=self._num_complex_levels, custom_constraint=self.complexity_level_greedy_search_custom_constraint)
            levels = result.levels
            rows = result.metric_rows(self.complexity_metric_list, self.complexity_level_greedy_search_complexity_metric,
                                      self.complexity_level_greedy_search_performance_metric)
            self.complexity_search_result = result
        dev = self.device
        self._complexity_param_all_levels = nn.ModuleList(
            [ParamDictModuleWrapper({n: self.node_generators[n](v) for n, v in lvl.items()}) for lvl in levels]).to(dev)
        self._num_complex_levels = len(levels)
        if rows is not None and hasattr(self, "_complexity_metric_list_cache"):
            self._complexity_metric_list_cache = torch.tensor(rows, dtype=torch.float32, device=dev)
        self._complexity_param_valid.fill_(True)
        self._valid_host = None
        self.eval()

    def set_task(self, task, *args, **kwargs) -> bool:
        if self._num_tasks == 0:
            return False
        self._current_task_idx = self.task_names.index(task) if task in self.task_names else int(task)
        return True

    @property
    def num_tasks(self):
        return max(1, self._num_tasks)
