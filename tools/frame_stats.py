import sys, json, numpy as np, torch
sys.path.insert(0,'/root/repo')
from gaussiancity_amd import _native as N, ext, synth
from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper
dev=torch.device('cuda:0')
for cname in ('C3','C2'):
    cfg,sc=synth.make_scene(cname)
    W,H,P=cfg['W'],cfg['H'],cfg['P']
    wr=GaussianRasterizerWrapper(synth.intrinsics(W,H),(W,H),device=dev)
    t={k:torch.from_numpy(v).to(dev) for k,v in sc.items() if isinstance(v,np.ndarray)}
    e=torch.Tensor([])
    for pose in (0,6):
        pos,quat=synth.orbit_poses()[pose]
        rs=wr._get_gaussian_rasterization_settings(pos,quat)._replace(sh_degree=3)
        o=ext.rasterize_gaussians(rs.bg,t['means3D'],e,t['opacities'],t['scales'],t['rotations'],1.0,e,rs.view_matrix,rs.proj_matrix,rs.tanfovx,rs.tanfovy,H,W,t['shs'],3,rs.campos,False,False)
        R,_,radii,_,_,img=o
        L=N.get_layout(P,W,H,R)
        T=((W+15)//16)*((H+15)//16)
        ranges=img[L.img_ranges:L.img_ranges+8*T].view(torch.int32).view(T,2).cpu().numpy()
        nc=img[L.img_n_contrib:L.img_n_contrib+4*W*H].view(torch.int32).view(H,W).cpu().numpy()
        ln=ranges[:,1]-ranges[:,0]
        rad=radii.cpu().numpy(); rv=rad[rad>0]
        print(cname,'pose',pose,'R',R,'vis',len(rv),'tile len mean %.0f p50 %d p90 %d p99 %d max %d'%(ln.mean(),*np.percentile(ln,[50,90,99]),ln.max()),
              '| n_contrib mean %.1f p50 %d p99 %d max %d'%(nc.mean(),*np.percentile(nc,[50,99]),nc.max()),'| radius mean %.1f p50 %d p90 %d p99 %d max %d'%(rv.mean(),*np.percentile(rv,[50,90,99]),rv.max()), '| sum tilelen^1: %d, chunks: %d'%(ln.sum(), np.ceil(ln/256).sum()))
