"""Randomised cross-check of the fused Poisson operator (every kernel family behind dn_poisson_apply, random launch-plan overrides) against the
same loss composed from the single-launch operators behind autograd -- tools/fuzz_fused.py, a fixed seed."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_fused_operator_agrees_with_the_composed_loss_on_random_cases():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_fused.py")
    spec = importlib.util.spec_from_file_location("fuzz_fused", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    failures = mod.run(ncases=60, seed=11, verbose=False)
    assert not failures, "\n".join(failures)


def test_fused_fsdt_residuals_agree_with_the_composed_form_on_random_cases():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_fused.py")
    spec = importlib.util.spec_from_file_location("fuzz_fused", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    failures = mod.run_fsdt(ncases=40, seed=12, verbose=False)
    assert not failures, "\n".join(failures)
