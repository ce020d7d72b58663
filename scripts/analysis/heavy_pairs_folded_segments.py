"""Prototype (CPU, numpy, float64), the variant WITHOUT unfolding: segments of |x| fitted on the folded coordinates fail wherever a fold switch
falls inside a segment -- about one segment per view whatever their number (2 / 4 / 8 segments: 48 / 24 / 12 % of the fits), i.e. half of a pair's
range stays on the per-sample path with 4 segments.  (The 1.07e-5-bin errors are the reference's float Pi: the +-x sides differ by 2.78e-8 n_alpha,
which the product's fit carries as a separate constant.)  python scripts/analysis/heavy_pairs_folded_segments.py"""
import numpy as np, sys
sys.path.insert(0,'/root/repo')
import oracle
from epipolarconsistency_amd import synthetic
n,S,B=400,1024,768
Ps=synthetic.short_scan(n,S,S,0.308)
r_obj=oracle.object_radius(Ps[0],S,S)
D=np.sqrt(2.0)*S; step_t=D/B; range_t=B*step_t; num_samples=2*B*step_t
Cs=[oracle.source_position(P) for P in Ps]; PT=[oracle.pinvT(P) for P in Ps]
Pi_f=float(np.float32(3.14159265359))
def mapping(K,kappa,sgn):
    # float64 version of the reference mapping (float Pi constant), returns (xa, yd, fold)
    c,s=np.cos(kappa)*sgn,np.sin(kappa)
    K=K.astype(np.float64)
    l0=K[0]*c+K[3]*s; l1=K[1]*c+K[4]*s; l2=K[2]*c+K[5]*s
    a=np.arctan2(l1,l0)/Pi_f; a=np.where(a<0,a+2,a)
    d=-(l2/np.hypot(l0,l1))/range_t+0.5
    fold=a>1; a=np.where(fold,a-1,a); d=np.where(fold,1-d,d)
    return a*B+0.5, d*B+0.5, fold
def cheb_nodes(m): return np.cos(np.pi*(np.arange(m)+0.5)/m)
def fit_seg(K,kmax,xa_,xb_,coord,degE=5,degO=4):
    za,zb=xa_**2,xb_**2; zc,zh=(za+zb)/2,(zb-za)/2
    w=cheb_nodes(6); z=zc+zh*w; x=np.sqrt(z); kap=x*kmax
    p=mapping(K,kap,+1); m=mapping(K,kap,-1)
    if not (np.all(p[2]==p[2][0]) and np.all(m[2]==m[2][0])): return None
    fp,fm=p[coord],m[coord]
    E=(fp+fm)/2; O=(fp-fm)/(2*x)
    cE=np.polynomial.chebyshev.chebfit(w,E,degE); cO=np.polynomial.chebyshev.chebfit(w,O,degO)
    # dense check
    xs=np.linspace(xa_,xb_,400); ws=(xs**2-zc)/zh
    pe=mapping(K,xs*kmax,+1); me=mapping(K,xs*kmax,-1)
    if not (np.all(pe[2]==p[2][0]) and np.all(me[2]==m[2][0])): return None
    Ev=np.polynomial.chebyshev.chebval(ws,cE); Ov=np.polynomial.chebyshev.chebval(ws,cO)
    err=max(np.abs(Ev+xs*Ov-pe[coord]).max(), np.abs(Ev-xs*Ov-me[coord]).max())
    return err
iu=np.triu_indices(n,1); d=iu[1]-iu[0]
heavy=np.flatnonzero((d>=325)&(d<=393))
rng=np.random.default_rng(0)
for nseg in (2,4,6,8):
    errs=[]; fails=0; tot=0
    for p in rng.choice(heavy,150,replace=False):
        i,j=iu[0][p],iu[1][p]
        K0,K1=oracle.computeK01(S/2,S/2,Cs[i],Cs[j],PT[i],PT[j],r_obj,num_samples)
        kmax=float(K1[7])
        # segment bounds aligned to 64-sample trips: 1448 samples -> 23 trips
        trips=23; bounds=np.round(np.linspace(0,trips,nseg+1)).astype(int)*64/1448.0; bounds[-1]=1.0; bounds[0]=1e-9
        for K in (K0,K1):
            for coord in (0,1):
                for s in range(nseg):
                    tot+=1
                    e=fit_seg(K,kmax,bounds[s],bounds[s+1],coord)
                    if e is None: fails+=1
                    else: errs.append(e)
    errs=np.array(errs)
    print("segments %d: fits %d, fold-switch failures %d (%.1f %%), error bins: median %.2e p99 %.2e max %.2e, share > 1e-5: %.2f %%"%(nseg,tot,fails,100*fails/tot,np.median(errs),np.percentile(errs,99),errs.max(),100*(errs>1e-5).mean()))
