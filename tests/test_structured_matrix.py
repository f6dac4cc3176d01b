"""Parity on structured inputs as a MATRIX (run with -m gpu): ten input structures x seeds x weight gains, through Generator.forward on the
HIP path, against the CPU oracle in fp32 and in float64 (tools/structured_matrix.py holds the cases; the full 8-seed x 5-gain x 5-engine
matrix of the round is committed under profiles/r05/).

Why the criterion is what it is (profiles/r05/a_reference_self_consistency.txt, a_structured_matrix.txt; tools/precision_study.py): at
weight gain 2 (|Y| ~ 5-10, a trained model's output size) the decoder is ill-conditioned - wherever a channel's AdaIN gain 1 + gamma is near
zero, the instance norm that follows divides the rounding of everything before it by |1 + gamma| (net/transformer.py:108-113 then :49-56) -
and the fp32 REFERENCE, evaluated with another batch size or thread count, differs from ITSELF by up to 2e-3 (its own Generator module:
batch 1 against batch 48, 7e-4).  Over the round's full matrix (8 seeds x 10 cases at gain 2) the fp32 reference arithmetic is further than
1e-4 from its own float64 evaluation in 33 of 80 rows (median 7e-5, 90th percentile 8e-4, worst 6e-3), and |hip - oracle32| has the same
distribution (7e-5 / 7e-4 / 1e-2) on EVERY engine set: the distance to the fp32 reference measures the reference's rounding, not the HIP
path's.  "Within 1e-4 of the fp32 reference" has no single answer there; the float64 evaluation of the same function does.  The bounds:

  (1) gains up to 1.5 (|Y| <= 1; the fp32 reference is within 5e-7 of float64 everywhere): the LITERAL bar on every row,
      |hip - oracle32| < 1e-4 (measured: < 6e-7);
  (2) gain 2, as a population: the HIP path's error against float64 is no larger than the fp32 reference arithmetic's -
      median and 90th percentile of |hip - f64| <= those of |oracle32 - f64|, worst <= 2 x worst
      (full matrix: 1.3e-5 / 3.4e-4 / 5.6e-3 against 7.0e-5 / 7.7e-4 / 6.0e-3);
  (3) gain 2, per row: |hip - f64| <= max(1e-4 max(1, max|Y|), 2 |oracle32 - f64|) - two roundings of the same ill-conditioned map are
      samples of one heavy-tailed distribution, so a per-row ratio has no tight bound; 2 is what the data supports (full matrix: worst 1.49;
      round 4's arithmetic - literal AdaIN -> instance norm order, fp32 style MLP - reached 5.0, its rule allowed 4).

Gains 2.25 and 3 are in the committed matrix for the record only: |Y| reaches 80 and 2e4 there and fp32 itself falls apart (the reference is
5e-2 resp. 5e2 from float64).  This test runs a reduced matrix (gain 2: seeds 0-1; gain 1.5: seed 0; default engines) so that the GPU suite stays in minutes.
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(3000)
def test_structured_matrix_bounds():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    import structured_matrix as M
    rows, _ = M.run(seeds=2, gains=[2.0], n=24, out_path=None, engine_names=["default"])
    rows_lo, _ = M.run(seeds=1, gains=[1.5], n=24, out_path=None, engine_names=["default"])
    rows = rows + rows_lo
    M.print_tables(rows, M.summarise(rows))
    assert all(r["finite"] for r in rows)
    lo = [r for r in rows if r["gain"] <= 1.5]
    g2 = [r for r in rows if r["gain"] == 2.0]
    assert lo and g2
    worst_lo = max(r["e_ho"] for r in lo)
    assert worst_lo < 1e-4, f"(1) literal bar at gain <= 1.5: |hip - oracle32| = {worst_lo:.2e}"
    eh = np.array([r["e_h64"] for r in g2]); eo = np.array([r["e_o64"] for r in g2])
    print(f"[matrix] gain 2, {len(g2)} rows: |hip - f64| median {np.median(eh):.2e} p90 {np.percentile(eh, 90):.2e} max {eh.max():.2e}   "
          f"|oracle32 - f64| median {np.median(eo):.2e} p90 {np.percentile(eo, 90):.2e} max {eo.max():.2e}")
    assert np.median(eh) <= np.median(eo) and np.percentile(eh, 90) <= np.percentile(eo, 90) and eh.max() <= 2.0 * eo.max(), "(2) population bound"
    ratio = np.array([r["e_h64"] / max(1e-4 * max(1.0, r["ymax"]), 2.0 * r["e_o64"]) for r in g2])
    assert ratio.max() <= 1.0, f"(3) per-row bound exceeded by a factor {ratio.max():.2f}: {g2[int(ratio.argmax())]}"


@pytest.mark.timeout(3000)
def test_structured_matrix_bounds_two_plane_fp16_engine():
    """The same three bounds with mocha_set_option("gemm_f16x2", 1) - at 48 windows per case, where the path's GEMMs are batch-size launches
    and really run on the two-plane fp16 engine (24 windows run the few-rows kernels): gain 2, seed 0 and gain 1.5, seed 0, ten cases each;
    the larger matrix of the round is profiles/r05/x_structured_matrix_f16x2_48windows.txt."""
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    import structured_matrix as M
    rows, _ = M.run(seeds=1, gains=[2.0], n=48, out_path=None, engine_names=["f16x2"])
    rows_lo, _ = M.run(seeds=1, gains=[1.5], n=48, out_path=None, engine_names=["f16x2"])
    rows = rows + rows_lo
    assert all(r["finite"] for r in rows)
    lo = [r for r in rows if r["gain"] <= 1.5]; g2 = [r for r in rows if r["gain"] == 2.0]
    worst_lo = max(r["e_ho"] for r in lo)
    assert worst_lo < 1e-4, f"(1) literal bar at gain <= 1.5: |hip - oracle32| = {worst_lo:.2e}"
    eh = np.array([r["e_h64"] for r in g2]); eo = np.array([r["e_o64"] for r in g2])
    print(f"[matrix, f16x2] gain 2, {len(g2)} rows: |hip - f64| median {np.median(eh):.2e} p90 {np.percentile(eh, 90):.2e} max {eh.max():.2e}   "
          f"|oracle32 - f64| median {np.median(eo):.2e} p90 {np.percentile(eo, 90):.2e} max {eo.max():.2e}")
    assert np.median(eh) <= np.median(eo) and np.percentile(eh, 90) <= np.percentile(eo, 90) and eh.max() <= 2.0 * eo.max(), "(2) population bound"
    ratio = np.array([r["e_h64"] / max(1e-4 * max(1.0, r["ymax"]), 2.0 * r["e_o64"]) for r in g2])
    assert ratio.max() <= 1.0, f"(3) per-row bound exceeded by a factor {ratio.max():.2f}: {g2[int(ratio.argmax())]}"
