import sys; sys.path.insert(0,'pll-modules_amd')
import pllhip_ctypes as pc, numpy as np
lib=pc.PllLib(pc.PRODUCT_LIB)
for S,N in ((20,125000),(20,1000000),(61,25000),(10,200000)):
    with pc.build_instance(lib, states=S, rate_cats=4, ntips=6, nsites=N, coded=True) as a:
        pc.full_traversal(a); t=a.tree
        st=a.alloc_sumtable(); sa,sb=t.scaler_of(t.root_a),t.scaler_of(t.root_b)
        a.update_sumtable(t.root_a,t.root_b,sa,sb,st)
        import time
        try:
            a.newton_branch(sa,sb,st,0.1,1e-4,10.0,1e-5,64)
            t0=time.perf_counter(); n=0
            for _ in range(20):
                x,its,_t=a.newton_branch(sa,sb,st,0.1,1e-4,10.0,1e-5,64); n+=its
            dt=time.perf_counter()-t0
            print(S,N,"device loop:",x,its,"iterates", round(dt/n*1e6,2),"us per iterate")
            t0=time.perf_counter(); n=0
            for _ in range(20*its):
                a.derivatives(sa,sb,0.1+1e-6*_,st); n+=1
            dt=time.perf_counter()-t0
            print(S,N,"host calls:", round(dt/n*1e6,2),"us per derivative call (ctypes)")
        except RuntimeError as e: print(S,N,e)
