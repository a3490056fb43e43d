"""Complexity-level search of BaSIC -- the selection logic of
LatentGraphicalANSEntropyCoder.post_training_process (cbench/modules/entropy_coder/latent_graph.py:1397-1640).

Given the controller nodes (slim-width indices of g_a / h_a / h_s / g_s; index min_sample = most complex,
max_sample = least complex) and an evaluator ``index tuple -> (complexity, loss)``, pick one index tuple per
complexity level: for each target complexity the tuple with the lowest loss whose complexity does not exceed it.

Host logic only; the evaluator runs the codec on the MI355X (see LatentGraphicalANSEntropyCoder below in
latent_graph.py).  Behaviour pinned by tests/golden/complexity_search.npz, produced by the reference's own method
on synthetic (FLOPs, loss) tables.

Reference behaviours kept on purpose:
  * candidates are scanned in the order {least complex, most complex, then itertools.product in controller order}
    and a candidate replaces the incumbent when its loss is <= the incumbent's (ties: the LATER one wins) (:1582-1589);
  * the incumbent starts at the loss of the least complex tuple, so a target nobody meets raises ValueError (:1596-1597);
  * default targets are num_levels-2 points strictly between the two extremes, which are levels 0 and num_levels-1
    themselves (:1525-1529); with custom constraints the levels are exactly the picks (:1519-1524).
The reference's "iterative" variant indexes one past its target list on the last level (:1549) and cannot complete;
it is not offered here.
"""
import itertools
from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Sequence, Tuple


class ComplexitySearchResult:
    def __init__(self, names, levels, complexity, loss, table):
        self.names = list(names)
        self.levels: List[Dict[str, int]] = levels          # one {controller: index} per complexity level
        self.complexity: List[float] = complexity           # evaluator's complexity of each level
        self.loss: List[float] = loss
        self.table = table                                  # every evaluated tuple -> (complexity, loss)

    def metric_rows(self, metric_names: Sequence[str], complexity_metric: str, performance_metric: str):
        """Per level the row the reference stores in _complexity_metric_list_cache (:1603-1612)."""
        rows = []
        for lvl, c, l in zip(self.levels, self.complexity, self.loss):
            row = []
            for m in metric_names:
                if m == complexity_metric:
                    row.append(float(c))
                elif m == performance_metric:
                    row.append(float(l))
                elif m in lvl:
                    row.append(float(lvl[m]))
                else:
                    row.append(0.0)
            rows.append(row)
        return rows


def search_complexity_levels(evaluate: Callable[[Dict[str, int]], Tuple[float, float]],
                             names: Sequence[str], min_sample: Dict[str, int], max_sample: Dict[str, int],
                             num_levels: Optional[int] = None,
                             custom_constraint: Optional[Sequence[float]] = None) -> ComplexitySearchResult:
    names = list(names)
    most = {n: int(min_sample[n]) for n in names}    # "max_complexity_idx" (:1506)
    least = {n: int(max_sample[n]) for n in names}   # "min_complexity_idx" (:1507)
    key = lambda idx: tuple(idx[n] for n in names)

    c_least, l_least = evaluate(least)
    c_most, l_most = evaluate(most)
    if not (c_least < c_most and l_most < l_least):
        raise ValueError("Complexity should be configured as 0 max!")  # the reference's assert (:1512)

    table = OrderedDict()
    table[key(least)] = (c_least, l_least)
    table[key(most)] = (c_most, l_most)
    for tup in itertools.product(*[range(most[n], least[n] + 1) for n in names]):
        if tup not in table:
            table[tup] = tuple(evaluate(dict(zip(names, tup))))

    if custom_constraint:
        targets = [float(t) for t in custom_constraint]
    else:
        if num_levels is None or num_levels < 2:
            raise ValueError("num_levels >= 2 or custom_constraint is required")
        targets = [c_most - i / (num_levels - 1) * (c_most - c_least) for i in range(1, num_levels - 1)]

    picks = []
    for target in targets:
        best, best_loss = None, l_least
        for tup, (c, l) in table.items():
            if c <= target and l <= best_loss:
                best, best_loss = tup, l
        if best is None:
            raise ValueError(f"no controller setting meets complexity target {target}")
        picks.append(best)

    order = picks if custom_constraint else [key(most)] + picks + [key(least)]
    return ComplexitySearchResult(names, [dict(zip(names, t)) for t in order], [table[t][0] for t in order],
                                  [table[t][1] for t in order], table)
