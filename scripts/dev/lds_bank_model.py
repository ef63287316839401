import sys, time, ctypes as C
import numpy as np
from tests import util
import positionbaseddynamics_amd as pbd
from positionbaseddynamics_amd import _ffi
lib=_ffi.lib
def run(spec,label,check=0):
    m=util.build_mine(spec)
    m.getConstraintGroups()
    for ba in (0,1):
        out=(C.c_uint64*8)()
        t=time.time()
        r=lib.pbdx_debug_plan_lds_model(m._h, ba, check, out)
        dt=time.time()-t
        if r: print('ERR', pbd._ffi.lib.pbdx_last_error()); continue
        o=list(out)
        print('%-22s bank_aware=%d: reads %9d groups -> %9d cycles (x%.2f)  writes %9d -> %9d (x%.2f)  table %9d -> %9d (x%.2f)  slots %d  plan %.2fs wall %.1fs'%(
            label,ba,o[0],o[1],o[1]/max(o[0],1),o[2],o[3],o[3]/max(o[2],1),o[4],o[5],o[5]/max(o[4],1),o[6],o[7]/1e6,dt))
n=int(sys.argv[1]) if len(sys.argv)>1 else 300
run(util.cloth_spec(n,n,4,3),'cloth %dx%d'%(n,n),check=int(sys.argv[2]) if len(sys.argv)>2 else 0)
