#!/usr/bin/env python
"""Random-shape campaign of the -m gpu kernel test bodies on the host emulation (no GPU): ragged attention windows for head_dim 16 /
17..64 / RPE, segmented reductions, pad and attention tables, norms, Linear with odd channel counts, convolution shape / dtype
combinations.   python tools/emu_fuzz.py <seed> <seconds> [conv|conv6|wgrad3]      (under tools/emu_asan.sh-style ASAN: see that script)
`conv6`: random gather tables (density 0 .. 1, 2 .. 27 table rows, 1 .. 700 rows) through conv5 and conv6 (PTC_CONV6=2): bit-identical; `wgrad3`: the same for wgrad2 / wgrad3.
Round 2: 2199 + 15 cases without a failure; 1 x 300 s under AddressSanitizer without a report; conv6: 2041 cases, wgrad3: 837 + 1550 cases (the second campaign, c_in 32 / 64, found a round-capacity error: fixed)."""
import os, sys, time, torch, random, numpy as np, faulthandler; faulthandler.enable()
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tests"))
import emu_backend
emu_backend.build()
import test_gpu_kernels as T
dev=torch.device("cpu")
rnd=random.Random(int(sys.argv[1]) if len(sys.argv)>1 else 0)
fails=[]
def run(name, **kw):
    try:
        with emu_backend.emulated_ops():
            getattr(T,name)(dev, **kw)
        return True
    except BaseException as e:
        fails.append((name,kw,type(e).__name__,str(e)[:200])); print("FAIL",name,kw,type(e).__name__,str(e)[:200],flush=True); return False
def conv6_case():
    from pointcept_amd import ops
    kv=rnd.choice([2,3,8,9,27]); n_out=rnd.choice([1,15,16,17,31,32,33,127,128,129,300,700]); n_in=rnd.choice([1,5,n_out,2*n_out+3])
    cin=rnd.choice([32,64]); cout=rnd.choice([32,64,96,128]); dt=rnd.choice([torch.bfloat16,torch.float16]); p=rnd.choice([0.0,0.03,0.2,0.35,0.5,0.8,1.0])
    g=torch.Generator().manual_seed(rnd.randint(0,1<<30))
    nbr=torch.randint(0,n_in,(kv,n_out),generator=g,dtype=torch.int32)
    nbr=torch.where(torch.rand(kv,n_out,generator=g)<p, nbr, torch.full_like(nbr,-1))
    x=torch.randn(n_in,cin,generator=g).to(dt); w=(torch.randn(cout,kv,cin,generator=g)*0.1).to(dt); b=torch.randn(cout,generator=g)
    kw=dict(kv=kv,n_out=n_out,n_in=n_in,cin=cin,cout=cout,dt=dt,p=p)
    try:
        with emu_backend.emulated_ops():
            os.environ.pop("PTC_CONV6",None); a=ops.spconv_fwd(x,w,b,nbr)
            os.environ["PTC_CONV6"]="2"; c=ops.spconv_fwd(x,w,b,nbr); os.environ.pop("PTC_CONV6",None)
        ref=b.view(1,-1)+sum(torch.where((nbr[k]>=0).view(-1,1), x.float()[nbr[k].clamp(min=0).long()], torch.zeros(1))@w[:,k].float().t() for k in range(kv))
        assert torch.equal(a,c), f"conv6 != conv5: {float((a.float()-c.float()).abs().max())}"
        assert float((c.float()-ref).abs().max()) <= 2e-2*max(1.0,float(ref.abs().max())), "conv6 vs fp32 reference"
        return True
    except BaseException as e:
        os.environ.pop("PTC_CONV6",None)
        fails.append(("conv6",kw,type(e).__name__,str(e)[:200])); print("FAIL conv6",kw,type(e).__name__,str(e)[:200],flush=True); return False
def wgrad3_case():
    from pointcept_amd import ops
    kv=rnd.choice([1,2,3,8,9,27]); n_out=rnd.choice([1,15,31,32,33,63,64,65,127,129,300,700,1500]); n_in=rnd.choice([1,5,n_out,2*n_out+3])
    cin=rnd.choice([32,64]); cout=rnd.choice([16,32,40,48,64,128]); dt=rnd.choice([torch.bfloat16,torch.float16]); p=rnd.choice([0.0,0.03,0.2,0.35,0.5,0.8,1.0])
    g=torch.Generator().manual_seed(rnd.randint(0,1<<30))
    nbr=torch.randint(0,n_in,(kv,n_out),generator=g,dtype=torch.int32)
    nbr=torch.where(torch.rand(kv,n_out,generator=g)<p, nbr, torch.full_like(nbr,-1))
    x=torch.randn(n_in,cin,generator=g).to(dt); dy=torch.randn(n_out,cout,generator=g).to(dt)
    kw=dict(kv=kv,n_out=n_out,n_in=n_in,cin=cin,cout=cout,dt=dt,p=p)
    try:
        with emu_backend.emulated_ops():
            os.environ.pop("PTC_WGRAD3",None); a=ops.spconv_wgrad(x,dy,nbr)
            os.environ["PTC_WGRAD3"]="2"; c=ops.spconv_wgrad(x,dy,nbr); os.environ.pop("PTC_WGRAD3",None)
        assert torch.equal(a,c), f"wgrad3 != wgrad2: {float((a-c).abs().max())}"
        return True
    except BaseException as e:
        os.environ.pop("PTC_WGRAD3",None)
        fails.append(("wgrad3",kw,type(e).__name__,str(e)[:200])); print("FAIL wgrad3",kw,type(e).__name__,str(e)[:200],flush=True); return False
t0=time.time(); n=0
while time.time()-t0 < float(sys.argv[2]) if len(sys.argv)>2 else 120:
    k="conv" if len(sys.argv)>3 and sys.argv[3]=="conv" else rnd.choice(["attn","attn","hd","rpe","seg","pad","tables","addnorm","ln","linear"])
    if len(sys.argv)>3 and sys.argv[3]=="conv6":
        conv6_case(); n+=1; continue
    if len(sys.argv)>3 and sys.argv[3]=="wgrad3":
        wgrad3_case(); n+=1; continue
    if k=="attn":
        lens=[rnd.choice([1,2,3,15,16,17,31,32,33,47,48,63,64,65,95,96,97,100,128,129,160,191,200]) for _ in range(rnd.randint(1,4))]
        run("test_attention_fwd_bwd", lens=lens, H=rnd.randint(1,5))
    elif k=="hd":
        D=rnd.choice([17,18,20,24,30,32,33,36,40,48,49,56,64])
        lim=1024 if D<=32 else 672 if D<=48 else 512
        lens=[min(lim,rnd.choice([1,2,31,32,33,64,65,100,129])) for _ in range(rnd.randint(1,3))]
        run("test_attention_other_head_dims_fwd_bwd", D=D, lens=lens, H=rnd.randint(1,4))
    elif k=="rpe":
        L=rnd.choice([16,33,48,64,100,128])
        run("test_attention_rpe_fwd_bwd", lens=[L]*rnd.randint(1,3), H=rnd.randint(1,4), bnd=rnd.choice([2,4,8,18,32]))
    elif k=="seg":
        run("test_segment_csr", dtype=rnd.choice([torch.float32,torch.bfloat16]), reduce=rnd.choice(["max","mean","sum","min"]), c=rnd.choice([1,3,5,16,33,64,100]))
    elif k=="pad":
        K=rnd.choice([1,2,3,4,16,48,128,1024])
        counts=[rnd.randint(1,4*K+3) for _ in range(rnd.randint(1,6))]
        run("test_patch_pad_maps", counts=counts, K=K)
    elif k=="tables":
        K=rnd.choice([4,16,48,128])
        counts=[rnd.randint(1,3*K+3) for _ in range(rnd.randint(1,4))]
        run("test_attn_tables_match_index_algebra", counts=counts, K=K)
    elif k=="addnorm":
        run("test_add_norm_fused_joint", c=rnd.choice([32,64,128,256,512]), mode=rnd.choice(["ln_add_ln","add_ln_scaled","add_cast","fp32"]))
    elif k=="ln":
        run("test_layer_norm_fwd_bwd", c=rnd.choice([32,64,128,256,512]), xdt=rnd.choice([torch.float32,torch.bfloat16]), ydt=rnd.choice([torch.float32,torch.bfloat16]))
    elif k=="conv":
        cin,cout,ks=rnd.choice([(8,32,5),(32,64,3),(64,32,3),(96,96,3),(48,96,3),(128,48,3),(16,16,3),(64,64,1),(32,96,1),(8,16,3)])
        run("test_spconv_fwd_and_wgrad", dtype=rnd.choice([torch.bfloat16,torch.float16,torch.float32]), cin=cin, cout=cout, ksize=ks)
    elif k=="linear":
        run("test_linear_identity_table", dtype=rnd.choice([torch.float32,torch.bfloat16]), n=rnd.choice([1,7,64,65,333,1000]), cin=rnd.choice([6,8,16,32,48,64,96]), cout=rnd.choice([8,16,20,32,64,96,128]))
    n+=1
print("cases",n,"fails",len(fails))
for f in fails: print(f)
