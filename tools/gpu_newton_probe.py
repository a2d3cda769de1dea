import sys; sys.path.insert(0,'pll-modules_amd')
import pllhip_ctypes as pc, numpy as np
lib=pc.PllLib(pc.PRODUCT_LIB)
for S,N in ((20,125000),(20,1000000),(61,25000),(10,200000)):
    with pc.build_instance(lib, states=S, rate_cats=4, ntips=6, nsites=N, coded=True) as a:
        pc.full_traversal(a); t=a.tree
        st=a.alloc_sumtable(); sa,sb=t.scaler_of(t.root_a),t.scaler_of(t.root_b)
        a.update_sumtable(t.root_a,t.root_b,sa,sb,st)
        try:
            print(S,N,a.newton_branch(sa,sb,st,0.1,1e-4,10.0,1e-5,32)[:2])
        except RuntimeError as e: print(S,N,e)
