"""CPU time the public path needs to ENQUEUE one training step (no synchronisation inside the timed loop) against the step's wall time.
usage: python tools/cpu_enqueue.py [users_per_step=32] [profile=1]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
args = bench.make_args(B, 'bf16')
dev = torch.device('cuda:0')
model, opt = bench.build_model(args, dev)
eng = model._engine()
g = torch.Generator().manual_seed(1); gc = torch.Generator().manual_seed(2)
content = bench.synth_content(65536, gc)
batches = [(i.to(dev), m.to(dev)) for i, m in bench.synth_batches(content, 65536, B, 2, g)]
def step(i):
    items, mask = batches[i % 2]
    eng.flat_g.zero_(); loss = eng.train_forward(items, mask); eng.train_backward(into_flat_grad=True); opt.step(grad_scale=1.0); return loss
for i in range(5): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20): step(i)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f'B = {B}: CPU enqueue {t_enq/20*1e3:.2f} ms/step, wall {t_all/20*1e3:.2f} ms/step')
if len(sys.argv) > 2 and sys.argv[2] == '0':
    sys.exit(0)
import cProfile, pstats
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(5): step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(14)
