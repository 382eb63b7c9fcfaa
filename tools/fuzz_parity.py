"""tools/fuzz_parity.py -- random shapes: every fused path against its one-launch-per-step twin on the GPU (and the oracle on
the small ones).  2D: the assembly ring loop vs fold + 24 step launches, the pre-normalised contract vs the raw one; 3D: persistent (plain,
folded, transposed, C channels on shared gates forward + backward) vs per-step / per-channel."""
import os, sys, random
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cspn_amd
from oracle import cspn2d_oracle, cspn3d_oracle



def forward2d_plan(g, h, s, N, norm, mode):
    """algo 'fused' with the assembly passes on plan `mode` (0 the product's linear plan, 1 the same without XCD-aware placement,
    2 the band groups of rounds 1-3) -- through the hook library, where the plan is an argument"""
    from cspn_amd import _lib
    if mode == 0:
        return cspn_amd.cspn2d_forward(g, h, s, N, norm, "fused")
    hooks = _lib.load_hooks()
    B, _, H, W = g.shape
    out = torch.empty_like(h)
    ws = torch.empty(max(1, cspn_amd.load().cspn2d_workspace_bytes(B, H, W, N)), dtype=torch.uint8, device=g.device)
    rc = hooks.cspn_debug_forward2d_plan(g.data_ptr(), h.data_ptr(), s.data_ptr() if s is not None else None, out.data_ptr(), B, H, W, N,
                                         _lib.NORM_TYPES[norm], mode, ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "cspn_debug_forward2d_plan")
    return out


def run(cases=40, seed=1, verbose=True):
  rnd = random.Random(seed)
  n2 = n3 = nb = nb2 = 0
  worst2 = 0.0
  for case in range(cases):
      # ---- 2D
      B, H = rnd.randint(1, 5), rnd.randint(1, 90)
      W = 4 * rnd.randint(64, 330)
      norm = rnd.choice(["8sum", "8sum_abs", "none"])
      sp = rnd.random() < 0.5
      N = rnd.choice([24, 24, 24, 48, 30, 12])
      gen = torch.Generator(device="cuda").manual_seed(case)
      g = torch.randn(B, 8, H, W, generator=gen, device="cuda")
      if norm == "none":
          g = g.abs() / (g.abs().sum(1, keepdim=True) + 0.2)
      h = torch.rand(B, 1, H, W, generator=gen, device="cuda") * 80
      s = (torch.rand(B, 1, H, W, generator=gen, device="cuda") < 0.02).float() * (h + 0.1) if sp else None
      b = cspn_amd.cspn2d_forward(g, h, s, N, norm, "stepwise")
      for loop in (0, 2, 8, 16, 18):   # the product's dispatch / the band-group plan; + 8: the 8 x 4 ring, + 16: the round-6 12 x 3 ring for every full first pass
          a = forward2d_plan(g, h, s, N, norm, loop)
          assert torch.equal(torch.isnan(a), torch.isnan(b)), ("2D nan", loop, B, H, W, norm, sp, N)
          err = float(((a - b).abs().nan_to_num()).max() / b.abs().nan_to_num().max().clamp_min(1e-30))
          worst2 = max(worst2, err)
          assert err <= 1e-5, ("2D", loop, B, H, W, norm, sp, N, err)
      if B * H * W <= 120000:
          r = cspn2d_oracle(g.cpu(), h.cpu(), None if s is None else s.cpu(), N, norm)
          e2 = float(np.nanmax(np.abs(a.cpu().numpy() - r)) / np.nanmax(np.abs(r)))
          assert e2 <= 1e-4, ("2D oracle", B, H, W, norm, sp, N, e2)
      if norm != "none":
          # round 5: the same call through the pre-normalised contract (cspn2d_normalize_f32 -> norm 'prenorm'): v_rcp x multiply vs an IEEE
          # division is the only difference
          wb = cspn_amd.cspn2d_normalize(g, norm)
          for algo in ("auto", "stepwise"):
              p_ = cspn_amd.cspn2d_forward(wb, h, s, N, "prenorm", algo)
              assert torch.equal(torch.isnan(p_), torch.isnan(b)), ("2D prenorm nan", algo, B, H, W, norm, sp, N)
              errp = float(((p_ - b).abs().nan_to_num()).max() / b.abs().nan_to_num().max().clamp_min(1e-30))
              worst2 = max(worst2, errp)
              assert errp <= 1e-5, ("2D prenorm", algo, B, H, W, norm, sp, N, errp)
      n2 += 1
      if N == 24 and norm != "none" and B * H * W <= 600000:
          # the backward of the same call (assembly sweeps + the recomputing final pass) against torch autograd through the plain-torch
          # restatement of the reference ops
          from tools.torch_path import cspn2d_torch
          go = torch.randn(B, 1, H, W, generator=gen, device="cuda")
          g0, h0 = g.clone().requires_grad_(True), h.clone().requires_grad_(True)
          cspn2d_torch(g0, h0, s, N, norm).backward(go)
          gg, gh = cspn_amd.cspn2d_backward(g, h, s, go, N, norm)
          for a_, r_ in ((gg, g0.grad), (gh, h0.grad)):
              fin = torch.isfinite(r_)
              assert torch.equal(torch.isfinite(a_), fin), ("2D bwd finite", B, H, W, norm, sp)
              # both sides are fp32: dG = dw / S - sign(G) T1 / S^2 cancels where S is small, which amplifies the rounding of either
              # (against a float64 oracle the worst element of such a case was 7.7e-6 of max|grad| for this path and 4.1e-6 for torch
              # itself, tools/fuzz_case22.py): the floor is the sum of the two noises
              tol = 2e-5 * float(r_[fin].abs().max()) + 2e-4 * r_[fin].abs()
              assert bool(((a_[fin] - r_[fin]).abs() <= tol).all()), ("2D bwd", B, H, W, norm, sp, float((a_[fin] - r_[fin]).abs().max()))
          nb2 += 1
          # round 6: the backward of the pre-normalised contract -- the gradient w.r.t. gate_wb (= w above) against torch autograd through the same loop
          from tools.torch_path import _gather8
          w0, h1 = wb.clone().requires_grad_(True), h.clone().requires_grad_(True)
          gs_, m_, cur = w0.sum(1, keepdim=True), (s.sign() if s is not None else None), h1
          for _ in range(N):
              cur = (1.0 - gs_) * h1 + (w0 * _gather8(cur)).sum(1, keepdim=True)
              if m_ is not None:
                  cur = (1.0 - m_) * cur + m_ * h1
          cur.backward(go)
          gw, gh2 = cspn_amd.cspn2d_backward(wb, h, s, go, N, "prenorm")
          for a_, r_ in ((gw, w0.grad), (gh2, h1.grad)):
              fin = torch.isfinite(r_)
              assert torch.equal(torch.isfinite(a_), fin), ("2D prenorm bwd finite", B, H, W, norm, sp)
              tol = 2e-5 * float(r_[fin].abs().max()) + 2e-4 * r_[fin].abs()
              assert bool(((a_[fin] - r_[fin]).abs() <= tol).all()), ("2D prenorm bwd", B, H, W, norm, sp, float((a_[fin] - r_[fin]).abs().max()))
      # ---- 3D
      B, D, H, W = rnd.randint(1, 3), rnd.randint(1, 40), rnd.randint(1, 70), 4 * rnd.randint(1, 60)
      N = rnd.randint(2, 14)
      norm = rnd.choice(["none", "none", "8sum_abs", "8sum"])
      sp = norm != "none" and rnd.random() < 0.5
      g = torch.randn(B, 26, D, H, W, generator=gen, device="cuda") if norm == "8sum" else torch.rand(B, 26, D, H, W, generator=gen, device="cuda")
      if norm == "none":
          g = g / g.sum(1, keepdim=True)
      h = torch.rand(B, 1, D, H, W, generator=gen, device="cuda")
      s = (torch.rand(B, 1, D, H, W, generator=gen, device="cuda") < 0.05).float() * (h + 0.1) if sp else None
      a = cspn_amd.cspn3d_forward(g, h, s, N, norm)              # auto: persistent where it takes the call
      b = cspn_amd.cspn3d_forward(g, h, s, N, norm, algo="stepwise")
      assert torch.equal(a, b), ("3D", B, D, H, W, N, norm, sp)
      n3 += 1
      if norm == "none" and B * D * H * W <= 400000:
          go = torch.randn(B, 1, D, H, W, generator=gen, device="cuda")
          gg, gf = cspn_amd.cspn3d_backward(g, h, go, N)          # fused sweeps where supported
          from oracle.backward import cspn3d_backward_oracle
          if B * D * H * W <= 60000:
              dG, dF = cspn3d_backward_oracle(g.cpu().numpy(), h.cpu().numpy(), go.cpu().numpy(), N)
              eg = float(np.abs(gg.cpu().numpy() - dG).max() / max(np.abs(dG).max(), 1e-30))
              ef = float(np.abs(gf.cpu().numpy() - dF).max() / max(np.abs(dF).max(), 1e-30))
              assert eg <= 2e-4 and ef <= 2e-4, ("3D bwd", B, D, H, W, N, eg, ef)
              nb += 1
          if case % 3 == 0 and B * D * H * W <= 200000:
              # round 5: C channels on shared gates, forward and backward in one call each, against the per-channel calls
              C = rnd.randint(2, 3)
              x = torch.rand(B, C, D, H, W, generator=gen, device="cuda")
              gox = torch.randn(B, C, D, H, W, generator=gen, device="cuda")
              with torch.no_grad():
                  fm = cspn_amd.affinity_propagate(x, g, 3, N)
              ggm, gfm = cspn_amd.cspn3d_backward_multi(g, x, gox, N)
              per = [cspn_amd.cspn3d_backward(g, x[:, c:c + 1].contiguous(), gox[:, c:c + 1].contiguous(), N) for c in range(C)]
              fw = torch.cat([cspn_amd.cspn3d_forward(g, x[:, c:c + 1].contiguous(), None, N, "none") for c in range(C)], 1)
              assert torch.equal(fm, fw), ("3D multi fwd", B, C, D, H, W, N)
              assert torch.equal(gfm, torch.cat([p_[1] for p_ in per], 1)), ("3D multi bwd feat", B, C, D, H, W, N)
              gs = sum(p_[0] for p_ in per)
              assert float((ggm - gs).abs().max()) <= 1e-5 * float(gs.abs().max().clamp_min(1e-30)), ("3D multi bwd gate", B, C, D, H, W, N)
  msg = "FUZZ OK: %d 2D cases (worst fused-vs-stepwise rel diff %.3g), %d 2D backward cases vs torch autograd, %d 3D cases bit-identical, %d 3D backward cases vs oracle" % (n2, worst2, nb2, n3, nb)
  if verbose:
      print(msg)
  return n2, nb2, n3, nb, worst2



if __name__ == "__main__":
    run(int(os.environ.get("FUZZ_CASES", "40")), int(os.environ.get("FUZZ_SEED", "1")))
