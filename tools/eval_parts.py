import sys, os, time, logging, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo') + '/tools')
import bench as B
from adapter4rec_amd import _lib as L
from adapter4rec_amd.data_utils import metrics as MT
dev = torch.device('cuda', 0)
args = B.make_args(32, 'bf16'); model, _ = B.build_model(args, dev); model.eval()
items, users = 65536, 32768
emb = torch.randn(items + 1, 64, device=dev)
rng = np.random.default_rng(1)
eval_seq, hist = {}, {}
for u in range(users):
    n = int(rng.integers(5, 22)); seq = [int(x) for x in rng.integers(1, items + 1, size=n)]
    eval_seq[u], hist[u] = seq, torch.LongTensor(seq[:-1])
uids = list(range(users))
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for step in (8192, 32768):
    os.environ['A4R_EVAL_USER_BATCH'] = str(step)
    print('step', step, 'eval_ranks ms', round(t(lambda: MT.eval_ranks(model, hist, eval_seq, emb, 512, args, uids)), 2))
P = MT._prepare_eval_set(eval_seq, hist, 21, dev)
inner = model
ub = torch.arange(users, device=dev)
print('gather emb ms', round(t(lambda: emb[P['ids'][ub].view(-1)].view(users, 20, 64)), 2))
ie = emb[P['ids'][ub].view(-1)].view(users, 20, 64)
print('user tower ms', round(t(lambda: inner.user_encoder(ie, P['mask'][ub], None)), 2))
