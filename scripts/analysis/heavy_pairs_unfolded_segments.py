"""Prototype (CPU, numpy, float64): can the kappa_max = pi/2 pairs of the BASELINE scan -- the 3.5 % on the pair kernel's per-sample path,
which cost 17 % of a launch (CHANGELOG.md, round 5) -- take the polynomial path?  Their sample coordinates UNFOLDED (continuous angle relative to
the line at kappa = 0, signed distance; the (alpha + pi, -t) fold applied per sample instead) fitted as E(z) + x O(z) (degree 5 / 4 in the shifted
z = x^2 of a segment of |x|, 6 Chebyshev nodes, segments aligned to 64-sample trips): 1 / 2 / 3 / 4 segments -> 20 / 48 / 78 / 100 % of the pairs
within 1e-5 bins (4 segments: median 8.6e-7, max 4.9e-6).  python scripts/analysis/heavy_pairs_unfolded_segments.py"""
import numpy as np, sys
sys.path.insert(0,'/root/repo')
import oracle
from epipolarconsistency_amd import synthetic
n,S,B=400,1024,768
Ps=synthetic.short_scan(n,S,S,0.308)
r_obj=oracle.object_radius(Ps[0],S,S)
D=np.sqrt(2.0)*S; step_t=D/B; range_t=B*step_t; num_samples=2*B*step_t
Cs=[oracle.source_position(P) for P in Ps]; PT=[oracle.pinvT(P) for P in Ps]
def unfolded(K,kappa):
    # continuous angle (in units of pi) and signed distance of the line of plane kappa (kappa in (-pi/2, pi/2)), relative to kappa = 0
    K=K.astype(np.float64)
    c,s=np.cos(kappa),np.sin(kappa)
    l0=K[0]*c+K[3]*s; l1=K[1]*c+K[4]*s; l2=K[2]*c+K[5]*s
    th0=np.arctan2(K[1],K[0])
    # angle relative to the line at kappa=0, continuous: atan2(cross, dot)
    dot=l0*K[0]+l1*K[1]; cross=K[0]*l1-K[1]*l0
    th=th0+np.arctan2(cross,dot)
    a=th/np.pi
    d=-(l2/np.hypot(l0,l1))/range_t+0.5
    return a*B, d*B   # unfolded texel-scale coordinates (no fold, no offsets)
def cheb_nodes(m): return np.cos(np.pi*(np.arange(m)+0.5)/m)
def fit_err(K,kmax,xa_,xb_,coord,degE=5,degO=4,nodes=6):
    za,zb=xa_**2,xb_**2; zc,zh=(za+zb)/2,(zb-za)/2
    w=cheb_nodes(nodes); z=zc+zh*w; x=np.sqrt(z); kap=x*kmax
    fp=unfolded(K,kap)[coord]; fm=unfolded(K,-kap)[coord]
    E=(fp+fm)/2; O=(fp-fm)/(2*x)
    cE=np.polynomial.chebyshev.chebfit(w,E,degE); cO=np.polynomial.chebyshev.chebfit(w,O,degO)
    xs=np.linspace(max(xa_,1e-6),xb_,600); ws=(xs**2-zc)/zh
    pe=unfolded(K,xs*kmax)[coord]; me=unfolded(K,-xs*kmax)[coord]
    Ev=np.polynomial.chebyshev.chebval(ws,cE); Ov=np.polynomial.chebyshev.chebval(ws,cO)
    return max(np.abs(Ev+xs*Ov-pe).max(), np.abs(Ev-xs*Ov-me).max())
iu=np.triu_indices(n,1); d=iu[1]-iu[0]
heavy=np.flatnonzero((d>=325)&(d<=393))
rng=np.random.default_rng(0)
sel=rng.choice(heavy,200,replace=False)
for nseg in (1,2,3,4):
    worst=[]
    for p in sel:
        i,j=iu[0][p],iu[1][p]
        K0,K1=oracle.computeK01(S/2,S/2,Cs[i],Cs[j],PT[i],PT[j],r_obj,num_samples)
        kmax=float(K1[7])
        trips=23; bounds=np.round(np.linspace(0,trips,nseg+1)).astype(int)*64/1448.0; bounds[-1]=1.0
        e=0
        for K in (K0,K1):
            for coord in (0,1):
                for s in range(nseg):
                    e=max(e,fit_err(K,kmax,bounds[s],bounds[s+1],coord))
        worst.append(e)
    worst=np.array(worst)
    print("unfolded, %d segment(s): per-pair worst error in bins: median %.2e p90 %.2e max %.2e; pairs within 1e-5: %.1f %%, within 1e-6: %.1f %%"%(nseg,np.median(worst),np.percentile(worst,90),worst.max(),100*(worst<1e-5).mean(),100*(worst<1e-6).mean()))
