#!/bin/bash
# Per-POSITION true durations of the one-window chain: rocprofv3 kernel trace of 300 eager streamed windows (fp32 bank, 16 384 rows),
# dispatches grouped by their position in the 40-launch sequence, with the gap to the previous kernel's end -> gpurun_out/chain_trace/chain.txt
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/chain_trace; rm -rf $out; mkdir -p $out
cat > $out/run.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mocha_sigasia2023_amd import ContextBank, Generator, StreamingCharacterizer, synthetic, synthetic_state_dict
dev = torch.device("cuda:0")
model = Generator(device=dev).load_state_dict(synthetic_state_dict(1777, 1.0)).eval()
g = torch.Generator(device=dev); g.manual_seed(2)
nm = torch.randn((16384, 90 * 256), device=dev, generator=g)
m_, s_ = synthetic.cnt_norm(7)
src = torch.from_numpy(synthetic.pose_windows(5, 8)).to(dev)
sc = StreamingCharacterizer(ContextBank(model, nm, nm.view(-1, 90, 256)), m_, s_, use_graph=False)
for i in range(20): sc.step(src[i % 8])
torch.cuda.synchronize()
sc.input.copy_(src[0][None]); torch.cuda.synchronize()
for i in range(300):
    sc.step()                    # zero-copy form: exactly the chain's kernels per window
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --output-format csv -d $out/raw -o t -- python3 $out/run.py > $out/stdout.txt 2> $out/stderr.txt
f=$(find $out/raw -name '*kernel_trace.csv' | head -1)
python3 - "$f" "$out/chain.txt" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the chain length: distance between the last two scans
scans = [i for i, n in enumerate(names) if "mocha_match_stream" in n]
L = scans[-1] - scans[-2]
tail = rows[scans[-1] - 299 * L - (scans[-1] % 1 ) : ]          # roughly the last 299 windows
start = scans[-200] - (scans[-200] - scans[-201])                  # align on a window: begin one chain before scan[-200]
# find the first kernel of a window: the one after the previous window's last kernel = position of scan minus its offset in the chain
first_after_scan = L - (scans[-1] - scans[-2])                     # 0
agg = collections.OrderedDict()
base = scans[-201]                                                 # a scan; positions are relative to it
sel = rows[base: base + 200 * L]
for i, r in enumerate(sel):
    pos = i % L
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    gap = (int(r["Start_Timestamp"]) - int(sel[i - 1]["End_Timestamp"])) / 1e3 if i else 0.0
    a = agg.setdefault(pos, [r["Kernel_Name"][:64], 0.0, 0.0, 0])
    a[1] += d; a[2] += gap; a[3] += 1
lines = [f"# one streamed window = {L} kernels (positions relative to the bank scan); average over 200 windows: duration, gap before the kernel (us)"]
td = tg = 0.0
for pos, (n, d, gp, c) in agg.items():
    lines.append(f"{pos:3d} {n:64s} {d / c:8.2f} {gp / c:7.2f}")
    td += d / c; tg += gp / c
lines.append(f"# sum of durations {td:.1f} us, sum of gaps {tg:.1f} us")
open(sys.argv[2], "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $out/raw
