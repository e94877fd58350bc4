"""Where do the ~36 __amd_rocclr_copyBuffer nodes of a replayed C2 step come from?  Counts Tensor.copy_ / clone / contiguous calls
that turn into a device-to-device memcpy (same dtype, both contiguous) while the step is captured, by call site."""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from pointcloudmatters_amd.bc import WORKLOADS, BCTrainer, build_act_policy, clone_batch, make_act_batch  # noqa: E402

dev = torch.device("cuda", 0)
wl = WORKLOADS["C2"]
torch.manual_seed(0)
pol = build_act_policy(pcd_npoints=wl["pcd_npoints"], sa_impl="fused").to(dev)
tr = BCTrainer(pol, total_steps=1000, precision="bf16", device=dev, mode="graph", optim=dict(accumulate_grad_batches=1))
batches = [make_act_batch(wl["batch"], wl["n_points"], seed=1000 + i, device=dev) for i in range(2)]
sites = collections.Counter()
orig_copy, orig_clone, orig_contig = torch.Tensor.copy_, torch.Tensor.clone, torch.Tensor.contiguous


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "pointcloudmatters_amd" in fr.filename or "bench" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
    return "?"


def copy_(self, src, *a, **k):
    if torch.cuda.is_current_stream_capturing() and torch.is_tensor(src) and src.is_cuda and self.is_cuda and src.dtype == self.dtype \
            and self.is_contiguous() and src.is_contiguous() and self.shape == src.shape:
        sites[("copy_", site(), tuple(self.shape))] += 1
    return orig_copy(self, src, *a, **k)


def clone(self, *a, **k):
    if self.is_cuda and torch.cuda.is_current_stream_capturing() and self.is_contiguous():
        sites[("clone", site(), tuple(self.shape))] += 1
    return orig_clone(self, *a, **k)


def contiguous(self, *a, **k):
    if self.is_cuda and torch.cuda.is_current_stream_capturing() and not self.is_contiguous():
        sites[("contiguous(copy kernel)", site(), tuple(self.shape))] += 1
    return orig_contig(self, *a, **k)


torch.Tensor.copy_, torch.Tensor.clone, torch.Tensor.contiguous = copy_, clone, contiguous
tr.training_step(clone_batch(batches[0]), prefetch=batches[1])
torch.Tensor.copy_, torch.Tensor.clone, torch.Tensor.contiguous = orig_copy, orig_clone, orig_contig
torch.cuda.synchronize()
tot = 0
for (kind, where, shape), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(f"{n:3d}  {kind:24s} {where:50s} {shape}")
    tot += n if kind != "contiguous(copy kernel)" else 0
print("memcpy-like calls seen from Python during capture:", tot, "(autograd-internal clones are not visible here)")
