import sys, torch
sys.path.insert(0, "/root/repo/tools/mb")
from mb_wgrad_batch import graph_time
dev="cuda"; R,E=800,512
bf=torch.bfloat16
dqkv=torch.randn(R,3,E,device=dev).to(bf)      # (B*L, 3, E): dq | dk | dv side by side
W=torch.randn(3*E,E,device=dev).to(bf)
dqk=dqkv[:, :2].reshape(R,2*E).contiguous(); dv=dqkv[:,2].contiguous()
def two():
    a=dqk@W[:2*E]; b=dv@W[2*E:]
def bmm3():
    A=torch.as_strided(dqkv,(3,R,E),(E,3*E,1)); torch.bmm(A, W.view(3,E,E))
def bmm3f():
    A=torch.as_strided(dqkv,(3,R,E),(E,3*E,1)); torch.bmm(A, W.view(3,E,E), out_dtype=torch.float32)
def one_sum():
    dqkv.view(R,3*E)@W
print({"two mm": graph_time(two), "bmm3 bf16": graph_time(bmm3), "bmm3 f32 out": graph_time(bmm3f), "one (sum only)": graph_time(one_sum)})
A=torch.as_strided(dqkv,(3,R,E),(E,3*E,1)); ref=torch.stack([dqkv[:,i].float()@W[i*E:(i+1)*E].float() for i in range(3)])
print((torch.bmm(A, W.view(3,E,E)).float()-ref).abs().max().item(), ref.abs().max().item())
