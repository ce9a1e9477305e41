import importlib,os,sys
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),"tests")]
pkg=importlib.import_module("gpu-homomorphic-encryption_amd")
from workload import rns_poly
n,L,w=8192,4,16
mod=pkg.find_ntt_primes(30,n,L)
def eng(coop, split):
    os.environ["FHE_HIP_COOP_POLYS"]=str(coop); os.environ["FHE_HIP_SPLIT_PAIRS_POLYS"]=str(split)
    try: return pkg.RnsNttEngine(n,mod)
    finally: os.environ.pop("FHE_HIP_COOP_POLYS"); os.environ.pop("FHE_HIP_SPLIT_PAIRS_POLYS")
for coop,split in ((0,0),(0,1000000),(64,0),(64,1000000)):
    e=eng(coop,split); K=e.relin_num_digits(w)
    keys=[[pkg.DeviceBuffer.from_numpy(rns_poly(7000+31*i+997*h,mod,n,1)) for i in range(L*K)] for h in range(2)]
    rk=e.import_relin_keys(w,keys[0],keys[1])
    for B in (1,4):
        ops=[pkg.DeviceBuffer.from_numpy(rns_poly(10+i,mod,n,B)) for i in range(4)]
        o0,o1=pkg.DeviceBuffer(B*L*n*32),pkg.DeviceBuffer(B*L*n*32)
        for _ in range(10): e.ct_multiply_relin(rk,o0,o1,ops[0],ops[1],ops[2],ops[3],B)
        pkg.capi.sync(); t=pkg.Timer(); t.start(e)
        for _ in range(200): e.ct_multiply_relin(rk,o0,o1,ops[0],ops[1],ops[2],ops[3],B)
        t.stop(e); pkg.capi.sync(); print("coop",coop,"split",split,"B",B, round(t.elapsed_ms()*1e3/200,1),"us", flush=True)
