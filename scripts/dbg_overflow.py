import sys; sys.path[:0]=['.','tests']
import numpy as np, ganon_amd as hip, ganon_fixtures as gf, gpu_util as gu
bins,rows,h=4096,4099,4
rng=np.random.default_rng(bins*31+h)
ibf=gf.random_ibf(bins,rows,h,0.4,seed=bins+h)
flt=hip.HipFilter.ibf(ibf.data,bins,rows,h)
seqs=[gu.random_seq(rng,int(L)) for L in [31,40,150,150,150,151,299,1000,20,4000]]
bases,off1,off2=gu.pack_reads(seqs,None)
for mm in (0, 100000):
    st=hip.HipStream(flt,len(seqs),bases.size,mm)
    st.submit(bases,off1,None,19,31,0.3)
    nh,status,mo,m=st.fetch()
    print("max_matches",mm,"nh",nh,"mo",mo,"len m",len(m), m[:3])
    nh,status,mo,m=st.fetch()
    print("  refetch mo",mo,"len m",len(m))
