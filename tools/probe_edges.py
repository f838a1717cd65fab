import os, sys
ROOT='/root/repo'
sys.path[:0]=[os.path.join(ROOT,'safe-offline-rl-with-diffusion-model_amd'), os.path.join(ROOT,'tests'), os.path.join(ROOT,'oracle')]
import numpy as np, torch
from trebi_amd import spec, synth
from trebi_amd.nets import TemporalUnet, ValueFunction
H,D=32,23
sd=synth.synth_state_dict(spec.unet_param_shapes(H,D,32,(1,2,4,8)),5)
net=TemporalUnet(H,D,0); net.load_state_dict({k:torch.from_numpy(v) for k,v in sd.items()}); net.set_max_timesteps(20); net=net.cuda()
for B in (0,):
    try:
        out=net(torch.zeros(B,H,D,device='cuda'),None,torch.zeros(B,dtype=torch.long,device='cuda'))
        print('unet B=0 ->',tuple(out.shape))
    except Exception as e: print('unet B=0 raised',type(e).__name__,str(e)[:150])
sdv=synth.synth_state_dict(spec.value_param_shapes(H,D,32,(1,2,4,8)),6)
v=ValueFunction(H,D,0); v.load_state_dict({k:torch.from_numpy(x) for k,x in sdv.items()}); v.set_max_timesteps(20); v=v.cuda()
try:
    y,g=v.value_and_input_grad(torch.zeros(0,H,D,device='cuda'),torch.zeros(0,dtype=torch.long,device='cuda'))
    print('value B=0 ->',tuple(y.shape),tuple(g.shape))
except Exception as e: print('value B=0 raised',type(e).__name__,str(e)[:150])
# out-of-range timestep
try:
    out=net(torch.zeros(2,H,D,device='cuda'),None,torch.tensor([25,-3],device='cuda'))
    torch.cuda.synchronize(); print('t out of range -> finite', bool(torch.isfinite(out).all()))
except Exception as e: print('t oor raised',type(e).__name__,str(e)[:150])
# wrong shape
try:
    net(torch.zeros(2,H,D+1,device='cuda'),None,torch.zeros(2,dtype=torch.long,device='cuda'))
except Exception as e: print('wrong D raised',type(e).__name__,str(e)[:120])
try:
    net(torch.zeros(2,H+1,D,device='cuda'),None,torch.zeros(2,dtype=torch.long,device='cuda'))
except Exception as e: print('wrong H raised',type(e).__name__,str(e)[:120])
from diffuser.models import GaussianDiffusion
from diffuser.models.diffusion import default_sample_fn
d=GaussianDiffusion(net,H,17,6,n_timesteps=20,loss_type='l2',clip_denoised=False,predict_epsilon=False).cuda()
try:
    s=d.p_sample_loop((0,H,D),{0:torch.zeros(0,17,device='cuda')},verbose=False,sample_fn=default_sample_fn)
    print('loop B=0 ->',tuple(s.trajectories.shape))
except Exception as e: print('loop B=0 raised',type(e).__name__,str(e)[:150])
