"""A short, seeded run of the randomised differential check (tools/fuzz_parity.py) per kind of input."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
fz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fz)

CASES = {"registration": 150, "fusion": 150, "subpixel": 80, "refine": 80, "fusion_wide": 60, "registration_big": 12,
         "apis": 150, "pipeline": 60, "registration_tiny": 400, "fusion_tiny": 300,
         "featureless": 300, "registration_wild": 200, "fusion_ff_extreme": 300, "mixed_radix": 100, "peaks10": 100,
         "window": 100, "consensus": 150, "fusion_chunked": 200, "u8": 150, "bitshuffle": 150, "subpixel_raw": 120, "pyramid_chunks": 150}


@pytest.mark.parametrize("kind", sorted(CASES))
def test_random_inputs_agree_with_the_oracle(kind):
    import torch
    assert torch.cuda.is_available()
    from image_stitcher_amd import fusion as fu, registration as reg, synth
    from oracle import fusion as ofu, refine as oref, registration as oreg, subpixel as osp
    rng = np.random.default_rng(1000 + sorted(CASES).index(kind))
    run = {"registration": lambda: fz.fuzz_registration(rng, reg, oreg, synth),
           "fusion": lambda: fz.fuzz_fusion(rng, fu, ofu, synth),
           "subpixel": lambda: fz.fuzz_subpixel(rng, reg, osp, synth),
           "subpixel_raw": lambda: fz.fuzz_subpixel_raw(rng, reg, osp, synth),
           "refine": lambda: fz.fuzz_refine(rng, reg, oref, synth),
           "fusion_wide": lambda: fz.fuzz_fusion_wide(rng, fu, ofu, synth),
           "registration_big": lambda: fz.fuzz_registration_big(rng, reg, oreg, synth),
           "apis": lambda: fz.fuzz_apis(rng, reg, fu, oreg, ofu, synth),
           "pipeline": lambda: fz.fuzz_pipeline(rng, reg, fu, synth),
           "registration_tiny": lambda: fz.fuzz_registration_tiny(rng, reg, oreg, synth),
           "fusion_tiny": lambda: fz.fuzz_fusion_tiny(rng, fu, ofu, synth),
           "featureless": lambda: fz.fuzz_featureless(rng, reg, oreg, synth),
           "registration_wild": lambda: fz.fuzz_registration_wild(rng, reg, oreg, synth),
           "fusion_ff_extreme": lambda: fz.fuzz_fusion_ff_extreme(rng, fu, ofu, synth),
           "mixed_radix": lambda: fz.fuzz_mixed_radix(rng, reg, oreg, synth),
           "peaks10": lambda: fz.fuzz_peaks10(rng, reg, oreg, synth),
           "window": lambda: fz.fuzz_window(rng, reg, oreg, synth),
           "consensus": lambda: fz.fuzz_consensus(rng, reg, oreg, synth),
           "fusion_chunked": lambda: fz.fuzz_fusion_chunked(rng, fu, ofu, synth),
           "u8": lambda: fz.fuzz_u8(rng, fu, ofu, synth),
           "bitshuffle": lambda: fz.fuzz_bitshuffle(rng, fu, ofu, synth),
           "pyramid_chunks": lambda: fz.fuzz_pyramid_chunks(rng, fu, ofu, synth)}[kind]
    for i in range(CASES[kind]):
        good, desc, _ = run()
        assert good, (i, desc)
