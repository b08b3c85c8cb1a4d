import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from gennbv_amd.env import synthetic as S
from gennbv_amd.env.config import TaskConfig
from gennbv_amd.env.state_encoding import OccupancyGridUpdater
n,g,h,w=256,64,240,320
dev="cuda:0"
cfg=TaskConfig(camera_width=w,camera_height=h,grid_size=g)
scene=S.make_scenes(n,g,seed=1,device=dev)
frames=S.make_frames(scene,cfg,4,seed=1,with_rgba=False)
upd=OccupancyGridUpdater(n,g,h,w,S.inverse_intrinsics(h,w),scene.range_gt,scene.voxel_size,scene.grid_gt,dev,max_steps_between_resets=100)
t8=torch.zeros(n,g**3,dtype=torch.int8,device=dev)
upd.self_clean=False
f=frames[0]
upd.update(f.depth_raw,f.seg_raw,S.c2w_from_view(f.view,scene.env_origins),f.poses.contiguous(),tri_i8_out=t8,fp32_out=False)
hit,path=upd.masks()
hit=hit.reshape(n,-1).cpu().numpy().astype(bool)
cnt=hit.sum(1)
e=int(cnt.argmax())
bits=hit[e].reshape(-1,32)   # words x 32
pw=bits.sum(1)               # popcount per word
print("env",e,"rays",cnt[e],"nonzero words",(pw>0).sum(),"max bits/word",pw.max(),"hist bits/word",np.bincount(pw)[:8],"...")
lane=pw.reshape(8,1024).sum(0)  # word wi -> lane wi%1024, 8 words per lane
print("bits per lane: max",lane.max(),"mean over nonzero",lane[lane>0].mean(),"lanes nonzero",(lane>0).sum())
wave=lane.reshape(16,64)
print("per wave max-lane bits:",wave.max(1),"per wave nonzero words:",(pw.reshape(8,16,64)>0).sum((0,2)))
em=int(np.argsort(cnt)[n//2]); pw2=hit[em].reshape(-1,32).sum(1); lane2=pw2.reshape(8,1024).sum(0)
print("median env rays",cnt[em],"nonzero words",(pw2>0).sum(),"max bits/lane",lane2.max())
