import sys, ctypes as C, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'skyfall-gs_amd'); sys.path.insert(0,'tests')
from sfgs import _lib as L
from sfgs.synth import scene
import diff_gauss as dg
lib=L.load()
dev=torch.device('cuda:0')
frame,g=scene(5000,130,77,seed=7,zrange=(2.,50.),scale_range=(0.01,2.0),mode='sh',sh_degree=1)
N=5000;W=130;H=77
t={k:(v.to(dev) if v is not None else None) for k,v in g.items()}
settings=dg.GaussianRasterizationSettings(H,W,frame['tanfovx'],frame['tanfovy'],0.1,None,frame['bg'].to(dev),1.0,frame['view'].to(dev),frame['proj'].to(dev),1,frame['campos'].to(dev),False,True)
keep=[]
fr=dg._frame(settings,dev,4,keep)
gs=L.SfgsGaussians(C.sizeof(L.SfgsGaussians),N,L.ptr(t['means3D']),L.ptr(t['scales']),L.ptr(t['rotations']),L.ptr(t['opacities']),None,L.ptr(t['shs']))
cap=80000
sizes=L.SfgsRasterSizes(C.sizeof(L.SfgsRasterSizes)); L.check(lib.sfgs_raster_sizes(N,W,H,cap,C.byref(sizes)))
u8=dict(dtype=torch.uint8,device=dev)
geom=torch.zeros(sizes.geom_bytes,**u8); tiles=torch.zeros(sizes.tiles_bytes,**u8); bins=torch.zeros(sizes.bins_bytes,**u8)
radii=torch.zeros(N,dtype=torch.int32,device=dev)
st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
L.check(lib.sfgs_raster_forward_plan(C.byref(fr),C.byref(gs),L.ptr(radii),L.ptr(geom),geom.numel(),L.ptr(tiles),tiles.numel(),L.ptr(bins),bins.numel(),cap,st))
cnt=L.SfgsRasterCounters(); L.check(lib.sfgs_raster_read_counters(L.ptr(tiles),C.byref(cnt),st))
D=cnt.num_duplicates; print('D',D,'ovf',cnt.overflow,'maxlist',cnt.max_tile_list,'nvis',cnt.num_visible)
T8=17*10
tb=tiles.cpu().numpy()
hdr=tb[:512].view(np.uint64); print('hdr',hdr[:6])
tc=tb[512:512+T8*4].view(np.uint32); print('tile_count sum',tc.sum(),'max',tc.max())
off=512+((T8*4+255)//256)*256
ts=tb[off:off+(T8+1)*4].view(np.uint32); print('tile_start last',ts[-1], 'monotone',(np.diff(ts.astype(np.int64))>=0).all(), (np.diff(ts.astype(np.int64))==tc).all())
stg=bins.cpu().numpy()[:D*16].view(np.uint32).reshape(D,4)
print('staging g range',stg[:,0].min(),stg[:,0].max(),'t range',stg[:,2].min(),stg[:,2].max(),'rank max',stg[:,3].max())
bad=(stg[:,2]>=T8)|(stg[:,3]>=tc[np.minimum(stg[:,2],T8-1)])
print('bad',bad.sum(), stg[bad][:10])
pos=ts[np.minimum(stg[:,2],T8-1)].astype(np.int64)+stg[:,3]
print('pos unique',len(np.unique(pos))==D, pos.max())
gv=geom.cpu().numpy()
dup=gv[((N*48+255)//256)*256:][:N*8].view(np.uint32).reshape(N,2)
print('dup count sum',dup[:,1].sum(), 'max n', dup[:,1].max())
per_g=np.bincount(stg[:,0],minlength=N)
mism=np.nonzero(per_g!=dup[:,1])[0]
print('mismatch gaussians',len(mism), mism[:10], per_g[mism[:10]], dup[mism[:10]])
zeros=(stg==0).all(axis=1).sum(); print('all-zero staging rows',zeros)
# check each gaussian's range is written with its own id
own=np.repeat(np.arange(N),dup[:,1]); idx=np.concatenate([np.arange(s,s+c) for s,c in dup if c>0])
print('slots holding foreign id', (stg[idx,0]!=own).sum())
import collections
bad_idx=idx[stg[idx,0]!=own][:10]; print(bad_idx, stg[bad_idx], own[stg[idx,0]!=own][:10])
allb=bins.cpu().numpy()[:cap*16].view(np.uint32).reshape(cap,4)
w=np.nonzero(allb[:,0]==54)[0]; print('rows with id 54:',w, allb[w])
nz=np.nonzero(allb.any(axis=1))[0]; print('nonzero rows', len(nz), 'max idx', nz.max())
w=np.nonzero(allb[:,0]==156)[0]; print('rows with id 156:',len(w), w[:20], dup[156])
