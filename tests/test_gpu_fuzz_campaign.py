"""A fixed handful of cases of the randomised differential campaign (tests/fuzz_campaign.py: random
model x kernel x usher x bias x dispatch override, native + device-sampled + replayed steps, GPU vs
oracle).  The campaign proper is run by hand with a time box; profiles/r04_fuzz_campaign.json holds
the summary of round 4's runs (7000+ cases, no mismatch), profiles/r05_fuzz_campaign.json round 5's (incl. the
`fast` profile: lean-multi, table, table-multi, Wang-Landau lean / multi shapes only)."""

import pytest

from tests import fuzz_campaign

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("profile,first", [("any", 11000003), ("lean", 12000006), ("fast", 14000009)])
def test_campaign_cases(profile, first):
    seen = {"ok": 0, "void": 0}
    kernels = set()
    for seed in range(first, first + 24):
        res = fuzz_campaign.run_case(seed, profile)
        assert res["status"] != "FAIL", res
        seen[res["status"]] += 1
        if res["status"] == "ok":
            kernels.add(res["desc"]["kernel_info"].split()[0])
    assert seen["ok"] >= 10 and len(kernels) >= 2, (seen, kernels)


def test_sampler_campaign_cases():
    """tests/fuzz_sampler.py: the same random cases through moca.Sampler (run continuing / restarting /
    after clear_samples, anneal, callable mod_update, bias arguments), mirrored call for call on the oracle."""
    from tests import fuzz_sampler

    ok, ops = 0, set()
    for seed in range(13500003, 13500003 + 20):
        res = fuzz_sampler.run_case(seed)
        assert res["status"] != "FAIL", res
        if res["status"] == "ok":
            ok += 1
            ops.update(op[0] for op in res["desc"]["ops"])
    assert ok >= 12 and {"run", "continue", "restart", "clear"} <= ops, (ok, ops)
