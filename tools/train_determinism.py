"""Development tool (GPU box): run-to-run spread of the training backward.  Two replicas (a: FusedAdamW with gradient slots, b:
torch.optim.AdamW) are stepped side by side and re-aligned after every step; per step prints the whole-gradient relative L2
difference a vs b, a vs a second backward of b, and b vs that second backward.  usage: python tools/train_determinism.py [bf16|f32]"""
import sys, copy, torch
sys.path.insert(0,'.')
from otpose_amd import synthetic as S
from otpose_amd.optim import FusedAdamW
from tests.test_gpu_train_slots import _pair, _targets, _loss, LR, WD, CLIP
cfg,a,b=_pair(sys.argv[1] if len(sys.argv) > 1 else "bf16")
x,margin=S.synthetic_clip(2,cfg.MODEL.IMAGE_SIZE); x,margin=x.cuda(),margin.cuda()
J,(w,h)=cfg.MODEL.NUM_JOINTS,cfg.MODEL.HEATMAP_SIZE
opt_a=FusedAdamW([p for p in a.parameters() if p.requires_grad],lr=LR,weight_decay=WD,max_grad_norm=CLIP)
opt_b=torch.optim.AdamW([p for p in b.parameters() if p.requires_grad],lr=LR,weight_decay=WD)
def rel(ga,gb):
    num=sum(float(((ga[n].double()-gb[n].double())**2).sum()) for n in ga); den=sum(float((gb[n].double()**2).sum()) for n in gb)
    return (num/den)**0.5
for it in range(6):
    g,wt=_targets(2,J,h,w,seed=11+5*it)
    opt_a.zero_grad(); opt_b.zero_grad()
    la=_loss(a,x,margin,g,wt); lb=_loss(b,x,margin,g,wt)
    la.backward(); lb.backward()
    torch.cuda.synchronize()
    opt_a.flat_grads()
    ga={n:p.grad.detach().clone() for n,p in a.named_parameters()}
    gb={n:(p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for n,p in b.named_parameters()}
    # b again (same weights): reference
    for p in b.parameters(): p.grad=None
    _loss(b,x,margin,g,wt).backward(); torch.cuda.synchronize()
    gr={n:(p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for n,p in b.named_parameters()}
    for n,p in b.named_parameters(): p.grad=gb[n]
    print('step',it,'a vs b %.2e  a vs b-again %.2e  b vs b-again %.2e'%(rel(ga,gb),rel(ga,gr),rel(gb,gr)))
    torch.nn.utils.clip_grad_norm_([p for p in b.parameters() if p.requires_grad],CLIP)
    opt_a.step(); opt_b.step()
    with torch.no_grad():
        for n,p in b.named_parameters(): p.copy_(dict(a.named_parameters())[n])
        for (_,ba),(_,bb) in zip(a.named_buffers(),b.named_buffers()): bb.copy_(ba)
