import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import libxaac_amd, oracle_lib
pseq, seq, cf = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(1); n = 8
spec, ovl = oracle_lib.random_case(rng, n, mag=17, ovl_mag=15)
ics = np.stack([np.full(n, seq), np.zeros(n)],1).astype(np.uint8); st = np.stack([np.full(n, pseq), np.zeros(n)],1).astype(np.uint8)
ctx = libxaac_amd.XaacContext(0, 0)
d = lambda a: torch.from_numpy(a).cuda()
t_ovl, t_st = d(ovl), d(st); pcm = torch.zeros(n*1024, dtype=torch.int16, device='cuda'); o32 = torch.zeros(n*1024, dtype=torch.int32, device='cuda')
ctx.imdct_process_batch(d(spec), d(ics), t_ovl, t_st, o32, pcm, None, ch_fac=cf)
torch.cuda.synchronize()
want = oracle_lib.load_oracle().imdct_batch(spec, ics, ovl, st, ch_fac=cf)
print(pseq, seq, cf, 'OK' if np.array_equal(pcm.cpu().numpy().reshape(n,1024), want['pcm16']) and np.array_equal(t_ovl.cpu().numpy(), want['overlap']) and np.array_equal(o32.cpu().numpy().reshape(n,1024), want['out32']) else 'MISMATCH')
