"""Randomised GPU-vs-oracle parity stress (run by hand on the GPU box: python tests/stress_gpu.py [seed] [families]).
Random families over group sizes, lengths, indel rates/lengths, protein/DNA, ls 1/3, tgapf 1/0.5, weighted or not;
every division of every family is aligned by the product and compared bit for bit with the CPU oracle (score, skeleton,
and the sum-of-pairs score of calcSpScore along the new path).  Kernel paths
can be forced with the G2G_* environment variables listed in DESIGN.md section 4."""
import sys, os, random, time
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
import numpy as np
import oraclelib
from prrn_aln_amd import engine, operator as op, sweep
from prrn_aln_amd.synth import make_family, DNA
ctx=engine.Context(); L=oraclelib.load()
rng=random.Random(int(sys.argv[1]) if len(sys.argv)>1 else 7)
tot=0; bad=0; t0=time.time()
for it in range(int(sys.argv[2]) if len(sys.argv)>2 else 14):
    n=rng.choice([2,3,4,5,6,9,14,20,33,48,64]); ln=rng.choice([60,120,250,420]); ind=rng.choice([0.005,0.02,0.05,0.09]); mi=rng.choice([3,8,20,45])
    dna=rng.random()<0.3; ls=rng.choice([1,1,3]); tg=rng.choice([1.0,1.0,0.5])
    kw=dict(n_seq=n,length=ln,seed=rng.randrange(10**6),indel=ind,max_indel=mi)
    if dna: kw['alphabet']=DNA
    fam=make_family(**kw)
    if min(len(s.replace('-','')) for s in fam.msa)==0: continue
    akw=dict(ls=ls,tgapf=tg)
    if dna: akw.update(molc=op.DNA,max_code=17)
    sw=sweep.Sweep(fam, op.AlnParam(**akw), weighted=rng.random()<0.7)
    ctx.reset_options()
    if it % 2: ctx.set_option("V6_MIN_STRIPS", 0)                           # every other family: v6 for _pf whatever the batch size
    res=op.align2_batch(ctx, sw.pwds)
    fs=op.calcSpScore_batch(ctx, sw.pwds, [skl for (_,skl,_) in res], stats=True)      # f1: sum-of-pairs score + FSTAT counters along the new path
    modes=set()
    for pw,(scr,skl,st),(val,gap,fst,raw,mch,mmc,unp) in zip(sw.pwds,res,fs):
        class H: c=pw.problem
        oscr,oc,otr=oraclelib.forward(L,H)
        ok = st==0 and scr==oscr and np.array_equal(skl, oraclelib.stdskl(L,otr))
        rc,oval,ogap,oraw,omch,ommc,ounp=oraclelib.spscore_stats(L,H,op.spparams(pw),skl)
        ok = ok and ((fst==0)==(rc==0)) and (rc!=0 or (val==oval and gap==ogap and raw==oraw and (mch,mmc,unp)==(omch,ommc,ounp)))
        tot+=1; bad+= (not ok); modes.add((pw.alnmode,pw.problem.noll))
        if not ok: print('MISMATCH', kw, akw, pw.alnmode, st, scr, oscr, fst, rc, val, oval)
    print(it, kw, akw, 'divisions', len(sw.pwds), 'modes', sorted(modes), 'bad so far', bad, flush=True)
print('total', tot, 'bad', bad, 'sec', round(time.time()-t0,1))
