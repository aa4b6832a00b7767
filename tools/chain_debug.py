"""Debug aid: a chain of 2 steps against the same steps one by one; dumps the chain's lr ring and where the tables first differ."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kge_keras_amd import _lib, models as M
C, lib = _lib.C, _lib.lib


def state(m):
    sd = torch.zeros(_lib.STEP_STATE_WORDS, dtype=torch.int64, device=m.device)
    o = m.optimizer
    od = _lib.OptDesc(_lib.OPT_IDS[o.kind], o.lr, o.beta_1, o.beta_2, o.epsilon, 1, None, None, None, None, sd.data_ptr())
    _lib.check(lib.kge_opt_state_init(C.byref(od), _lib.stream_ptr()))
    m._step_dev = sd


N, R, d = 301, 9, 24
rs = np.random.RandomState(21)
E0 = rs.uniform(-0.5, 0.5, (N, d)).astype(np.float32)
R0 = rs.uniform(-0.5, 0.5, (R, d)).astype(np.float32)
mk = lambda: M.DistMult(e_dim=d, r_dim=d, num_entities=N, num_relations=R, dp=0.0, init_entities=E0, init_relations=R0, loss_function="pointwize_logistic", margin=1.0)
n_pos, n_neg = 64, 1500
lib.kge_set_deterministic(1)
ref, chn = mk(), mk()
for m in (ref, chn):
    m.compile("adam")
    m._ensure_train_state(n_pos, n_neg)
    state(m)
ref._step_twins = (torch.empty_like(ref.E), torch.empty_like(ref.Rel))
cur = [chn.E, chn.Rel]
nxt = [torch.full_like(chn.E, 7.0), torch.full_like(chn.Rel, 7.0)]
md, ld = chn._model_desc(True), chn._loss_desc()
err = torch.zeros(1, dtype=torch.int32, device="cuda")
ws_f = chn._ws
print("ws bytes", ws_f.numel())
refs = []
for seq in range(3):
    pos = np.stack([rs.randint(0, N, n_pos), rs.randint(0, R, n_pos), rs.randint(0, N, n_pos)], 1).astype(np.int32)
    neg = np.stack([rs.randint(0, N, n_neg), rs.randint(0, R, n_neg), rs.randint(0, N, n_neg)], 1).astype(np.int32)
    ref.train_step(ref.plan_batch([pos, neg]))
    refs.append((ref.E.clone(), ref.Rel.clone(), ref._step_dev.clone()))
    plan = chn.plan_batch([pos, neg])
    od = chn._opt_desc()
    _lib.check(lib.kge_train_step_swap_chain(C.byref(md), C.byref(ld), C.byref(od), cur[0].data_ptr(), cur[1].data_ptr(), nxt[0].data_ptr(), nxt[1].data_ptr(),
                                             plan.buf.data_ptr(), plan.buf.numel(), n_pos, n_neg, chn._loss.data_ptr(), 0, chn._ws.data_ptr(), chn._ws.numel(), seq, _lib.stream_ptr()))
    torch.cuda.synchronize()
    last = (cur[1], nxt[1])
    cur, nxt = nxt, cur
    print("after launch", seq, "E diff vs ref step", seq, float((cur[0] - refs[seq][0]).abs().max()), "n", int((cur[0] != refs[seq][0]).sum()),
          "| state", chn._step_dev.tolist(), "ref", refs[seq][2].tolist())
    if seq:
        print("   Rel closed by this launch vs ref step", seq - 1, float((nxt[1] - refs[seq - 1][1]).abs().max()), "n", int((nxt[1] != refs[seq - 1][1]).sum()))
od = chn._opt_desc()
_lib.check(lib.kge_train_step_swap_flush(C.byref(md), C.byref(ld), C.byref(od), last[0].data_ptr(), last[1].data_ptr(), n_pos, n_neg, chn._loss.data_ptr(),
                                         chn._ws.data_ptr(), chn._ws.numel(), 2, err.data_ptr(), _lib.stream_ptr()))
torch.cuda.synchronize()
print("flush: Rel diff", float((cur[1] - ref.Rel).abs().max()), "state", chn._step_dev.tolist(), "ref", ref._step_dev.tolist(), "err", int(err.item()))
sd = ref._step_dev.view(torch.float32)
print("ref lr word", float(sd[2]))
# the lr ring lives in the workspace behind the pull header: find it by value
w = chn._ws.view(torch.float32)
for t in (1, 2, 3, 4):
    lr = np.float32(1e-3 * np.sqrt(1.0 - np.float64(np.float32(0.999)) ** t) / (1.0 - np.float64(np.float32(0.9)) ** t))
    idx = (w == float(lr)).nonzero().flatten().tolist()
    print("lr(t=%d) = %.9g found at float index" % (t, lr), idx[:6])
lib.kge_set_deterministic(0)
