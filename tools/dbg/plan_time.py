import torch, sys
sys.path.insert(0, "/root/repo")
from pointcloudmatters_amd.pointops import _common as C
import pointcloudmatters_amd.pointops as po
dev = torch.device("cuda:0")
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rows, n_dst, c, stride, off in [(1048576, 131072, 96, 99, 3), (393216, 65536, 96, 96, 0)]:
    idx = torch.randint(0, n_dst, (rows,), dtype=torch.int32, device=dev)
    src = torch.randn(rows if stride == 99 else rows // 3, stride, device=dev)
    dst = torch.empty(n_dst, c, device=dev)
    t_plan = timed(lambda: C.ScatterPlan(idx, n_dst))
    plan = C.ScatterPlan(idx, n_dst)
    t_sum = timed(lambda: C.segment_sum(dst, src, src_stride=stride, src_off=off, plan=plan, rowdiv=1 if stride == 99 else 3))
    print(rows, n_dst, "plan %.1f us  segsum %.1f us" % (t_plan, t_sum))
