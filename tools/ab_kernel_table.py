#!/usr/bin/env python
"""Per-kernel A/B table from rocprofv3 `--kernel-trace --stats` summaries of the SAME bench command on two builds of the library
(tools/round_start.sh: the shipped lib/ and lib_next/), next to a reference round's numbers:

    python tools/ab_kernel_table.py profiles/r04_bench_kernel_stats.csv gpurun_out/profiles_raw/r06/bench/*kernel_stats.csv \
                                    gpurun_out/profiles_raw/r06/bench_next/*kernel_stats.csv [--all] > profiles/r06_ab_kernels.md

Columns: average us per launch in the reference round | shipped | next, launches per profiled run, and the verdict per kernel
("next faster" needs >= 3 % and >= 0.1 us; the shipped column should reproduce the reference one -- same code object).  Only hand-written
kernels (pcm_*) unless --all.  Per FILE of csrc/ a summary line: a rewrite is adopted per file, only if no kernel of the file is slower."""
import csv
import re
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
show_all = "--all" in sys.argv
if len(args) != 3:
    sys.exit(__doc__)


def load(path):
    out = {}
    for r in csv.DictReader(open(path)):
        name = re.sub(r"void |\(anonymous namespace\)::", "", r["Name"])
        name = re.sub(r"\(.*\)$", "", name)  # drop the argument list
        out[name] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6)
    return out


ref, ship, nxt = (load(p) for p in args)
# kernel -> source file of csrc/ (by kernel-name prefix; good enough for a summary line)
FILES = (("pcm_fps", "fps"), ("pcm_knn", "knn"), ("pcm_drln", "drln"), ("pcm_ffn_ln", "ffn"), ("pcm_attn_small", "attn_small"),
         ("pcm_attn_flash", "attn_flash"), ("pcm_bn_", "bnrelu"), ("pcm_sa_", "sa_fused"), ("pcm_adamw", "optim"), ("pcm_xfer", "optim"),
         ("pcm_grad_sumsq", "optim"), ("pcm_add", "tokens"), ("pcm_colsum", "tokens"), ("pcm_reduce_batch", "tokens"), ("pcm_copy_batch", "tokens"),
         ("pcm_act_loss", "tokens"), ("pcm_cvae", "tokens"), ("pcm_coord_embed", "tokens"), ("pcm_incr", "tokens"))
NEXT_FILES = {"fps", "knn", "drln", "ffn", "attn_small", "attn_flash", "tokens", "optim", "sa_fused", "bnrelu"}


def file_of(k):
    for pre, f in FILES:
        if k.startswith(pre):
            return f
    return "-"


names = sorted(set(ship) | set(nxt), key=lambda k: -(ship.get(k, (0, 0, 0))[2]))
print("| kernel | file | launches | ref us | shipped us | next us | next vs shipped |")
print("|---|---|---|---|---|---|---|")
per_file = {}
tot = {"ship": 0.0, "next": 0.0}
for k in names:
    if not show_all and not k.startswith("pcm_"):
        continue
    c, s_us, s_ms = ship.get(k, (0, float("nan"), 0.0))
    _, n_us, n_ms = nxt.get(k, (0, float("nan"), 0.0))
    r_us = ref.get(k, (0, float("nan"), 0.0))[1]
    f = file_of(k)
    verdict = ""
    if s_us == s_us and n_us == n_us and f in NEXT_FILES:
        d = n_us - s_us
        verdict = "next faster" if (d <= -0.03 * s_us and d <= -0.1) else ("next SLOWER" if (d >= 0.03 * s_us and d >= 0.1) else "same")
        per_file.setdefault(f, []).append((k, verdict, s_ms, n_ms))
        tot["ship"] += s_ms
        tot["next"] += n_ms
    print("| `%s` | %s | %d | %.2f | %.2f | %.2f | %s |" % (k[:90], f, c, r_us, s_us, n_us, verdict))
print()
print("Per file (a rewrite is adopted only if no kernel of its file is slower on the same lease):")
for f in sorted(per_file):
    rows = per_file[f]
    slower = [k for k, v, _, _ in rows if v == "next SLOWER"]
    faster = [k for k, v, _, _ in rows if v == "next faster"]
    s, n = sum(r[2] for r in rows), sum(r[3] for r in rows)
    print("* `%s`: %d kernels, %d faster, %d slower; device time per profiled run %.3f ms -> %.3f ms: %s" % (
        f, len(rows), len(faster), len(slower), s, n, "KEEP FROZEN" if slower or not faster else "adopt next/%s.hip" % f))
print("\nAll rewritten files together: %.3f ms -> %.3f ms of device time per profiled run." % (tot["ship"], tot["next"]))
