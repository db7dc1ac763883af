"""The optimised code paths against their general forms: one full-width denoising step (2 views, CFG) run in subprocesses
under the engine's environment switches.  Every switch selects a mathematically equivalent computation (other
split-K / tile plan, biases folded into the accumulators or added in the epilogue, the 2-D encoder as one kernel or layer by
layer, ResBlock conv1's slabs summed by the reduce pass or by GroupNorm 2); they differ in fp32 summation order only.

What that costs is itself a property worth pinning: a 1e-7 perturbation changes every GroupNorm's statistics, tips a few fp16
operand roundings in the next layer, those tip more, and within a handful of layers the rounding noise of the two runs is
uncorrelated.  So two equivalent paths agree to the SAME level as either agrees with the fp32 oracle (measured: guided eps
4.8e-4 ... 5.6e-4, x_prev 2.4e-5 ... 2.8e-5) -- not to 1e-6 -- while the default paths are bit-reproducible from process to
process.  The bound is therefore the parity budget, 1e-3."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

BOUND = 1e-3
VARIANTS = {
    "general_row_mapping": {"MVD_NO_PLAIN": "1"},
    "k_step_plan": {"MVD_OLD_PLAN": "1"},
    "layered_encoder_own_reduce": {"MVD_NO_FUSED_ENC": "1", "MVD_NO_DEFER_REDUCE": "1"},
    "scalar_layernorm_f32_bn128": {"MVD_LN_SCALAR": "1", "MVD_IGEMM_F32_BN128": "1"},
    # the row-chain kernel (k_rowchain.hip: transformer tail in one launch) at a batch where the default is the layered GEMMs,
    # together with the halo kernel of rounds 1-4 in place of conv3x (k_conv3x.hip) for the 3x3 convolutions
    # (the row-head kernel -- proj_in, LayerNorm1, q|k|v in one launch -- rides on the same switch)
    "rowchain_at_every_batch_halo_conv": {"MVD_ROWCHAIN_MIN_ROWS": "0", "MVD_NO_CONV3X": "1"},
    # round 6, the restructured paths against their round-5 forms (two processes instead of one per switch: each is a full-width
    # model load).  (a) LayerNorm1 / LayerNorm3 sum the split-K slabs of proj_in / to_out (off by default: measured neutral), the
    # DepthTransformer's GroupNorms do NOT sum theirs (the folded FF2 + proj_out GEMM then takes its fp16 t2 from the slab form);
    # and the ResBlocks' skip convolutions on a helper stream beside GroupNorm1 -> conv1 -> GroupNorm2 (opt-in: measured neutral)
    "ln_sums_slabs_cond_reduces_skip_conv_on_helper_stream": {"MVD_LN_DEFER": "1", "MVD_NO_COND_DEFER": "1", "MVD_SKIP_SIDE": "1"},
    # (b) every DepthTransformer folds its own context projection and runs over ALL samples (no per-level stacked GEMM, no cached
    # constant image for the context-free half), FF2 and proj_out as two GEMMs with the fp16 intermediate between them, the 4 -> 8
    # Upsample as the 9-tap GEMM on the fp32 source, the step's head on the UNet's stream
    "round5_forms_of_the_round6_restructurings": {"MVD_NO_CTX_GROUP": "1", "MVD_NO_COND_CONST": "1", "MVD_NO_FFP": "1",
                                                  "MVD_NO_UP_CONV3X": "1", "MVD_HEAD_ON_MAIN": "1"},
}


def _run(tmp_path, name, env_extra):
    out = tmp_path / f"{name}.pt"
    env = dict(os.environ, **env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "variant_step.py"), str(out)], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    return torch.load(out)


def test_optimised_paths_equal_their_general_forms(tmp_path):
    ref = _run(tmp_path, "default", {})
    again = _run(tmp_path, "default_again", {})
    assert torch.equal(again["eps"], ref["eps"]), "the default paths are not bit-reproducible across processes"
    worst = {}
    for name, env_extra in VARIANTS.items():
        got = _run(tmp_path, name, env_extra)
        for key in ("eps", "x_prev"):
            rel = ((got[key] - ref[key]).norm() / ref[key].norm()).item()
            print(f"[property] {name}: {key} vs default build paths relL2 = {rel:.2e}")
            worst[(name, key)] = rel
    assert max(worst.values()) < BOUND, worst
