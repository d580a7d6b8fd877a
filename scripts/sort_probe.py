"""Times rdg_sort_pairs (stable LSD radix sort, 8-bit passes) through the C-ABI and checks it against torch's stable sort.
usage: sort_probe.py <n> <key bits> [<value range of the keys' low digit: 'tile' = the dense frame's tile-id pattern>]"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from rodygs_amd.rigidity import _sort_by_key
n, bits = int(sys.argv[1]), int(sys.argv[2])
g = torch.Generator().manual_seed(1)
keys = torch.randint(0, 1 << min(bits, 62), (n,), generator=g, dtype=torch.int64).cuda()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(4):
    ev[0].record(); srt, idx = _sort_by_key(keys, bits); ev[1].record(); torch.cuda.synchronize()
ref, ridx = torch.sort(keys, stable=True)
ok = bool(torch.equal(srt, ref) and torch.equal(idx, ridx))
print(f"n={n} bits={bits}: {ev[0].elapsed_time(ev[1]):.3f} ms incl. glue, equal to torch's stable sort: {ok}")
