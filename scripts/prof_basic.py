import sys, os, time, torch
sys.path.insert(0, '/root/repo')
from scripts.ar_bench import prep
from cbench_basic_amd.presets import basic_codec
c = prep(basic_codec()); c.set_complex_level(7)
x = torch.rand(64,3,256,256).cuda()
d = c.compress(x); c.decompress(d); torch.cuda.synchronize()
c.entropy_coder.collect_profiler_results()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
t0=time.time(); d = c.compress(x); torch.cuda.synchronize(); t1=time.time()
pr.disable()
print("compress", (t1-t0)*1e3, "ms")
for k,v in sorted(c.entropy_coder.collect_profiler_results().items(), key=lambda kv:-kv[1])[:10]: print(f"{k:60s} {v*1e3:8.1f} ms")
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
