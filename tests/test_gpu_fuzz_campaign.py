"""A fixed handful of cases of the randomised differential campaign (tests/fuzz_campaign.py: random
model x kernel x usher x bias x dispatch override, native + device-sampled + replayed steps, GPU vs
oracle).  The campaign proper is run by hand with a time box; profiles/r04_fuzz_campaign.json holds
the summary of the round's runs (7000+ cases, no mismatch)."""

import pytest

from tests import fuzz_campaign

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("profile,first", [("any", 11000003), ("lean", 12000006)])
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
