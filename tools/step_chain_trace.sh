#!/bin/bash
# Per-POSITION true durations and gaps of the demo step (585 + 585 windows, characterize_pair): rocprofv3 kernel trace of 24 steps,
# dispatches grouped by their position in the step's launch sequence -> gpurun_out/step_trace/chain.txt
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/step_trace; rm -rf $out; mkdir -p $out
cat > $out/run.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mocha_sigasia2023_amd import Generator, synthetic, synthetic_state_dict
dev = torch.device("cuda:0")
model = Generator(layout="mixamo", device=dev).load_state_dict(synthetic_state_dict(1777, 1.0, "mixamo")).eval()
W = int(os.environ.get("MOCHA_TRACE_WINDOWS", "585"))      # windows per clip
src = torch.from_numpy(synthetic.pose_windows(1777, W, 22)).to(dev)
cha = torch.from_numpy(synthetic.pose_windows(4242, W, 22)).to(dev)
mean, std = (torch.from_numpy(a).to(dev) for a in synthetic.cnt_norm(7))      # device tensors: numpy statistics would be copied (and the
                                                                              # device drained) on every call
for _ in range(24): model.characterize_pair(src, cha, mean, std)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --output-format csv -d $out/raw -o t -- python3 $out/run.py > $out/stdout.txt 2> $out/stderr.txt
f=$(find $out/raw -name '*kernel_trace.csv' | head -1)
python3 - "$f" "$out/chain.txt" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "mocha_match_select" in r["Kernel_Name"]]
L = marks[-1] - marks[-2]
base = marks[-17]
sel = rows[base: base + 16 * L]
agg = collections.OrderedDict()
for i, r in enumerate(sel):
    pos = i % L
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    gap = (int(r["Start_Timestamp"]) - int(sel[i - 1]["End_Timestamp"])) / 1e3 if i else 0.0
    a = agg.setdefault(pos, [r["Kernel_Name"][:70], 0.0, 0.0, 0])
    a[1] += d; a[2] += gap; a[3] += 1
lines = [f"# one demo step = {L} kernels (positions relative to mocha_match_select); average over 16 steps: duration, gap before the kernel (us)"]
td = tg = 0.0
for pos, (n, d, gp, c) in agg.items():
    lines.append(f"{pos:3d} {n:70s} {d / c:9.2f} {gp / c:7.2f}")
    td += d / c; tg += gp / c
lines.append(f"# sum of durations {td:.1f} us, sum of gaps {tg:.1f} us")
open(sys.argv[2], "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $out/raw
