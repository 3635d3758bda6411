"""Which kernel family a batch call takes is decided per batch, not only at create time: a row too long for the fold kernel's
32-bit byte offsets must leave the fused route when the kernel that would take it instead does not hold the contract
(log / dB columns on the pass-scaled kernels) or does not exist for the plan (hops only the fold kernel's ring holds)."""
import numpy as np
import pytest

import spotcheck
import util
from syllable_detector_swift_amd import SyllableDetector, _abi, nets, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", ["db", "hop176"])
def test_rows_past_2_gib_leave_the_fused_route_where_only_the_fold_kernel_holds_the_contract(oracle_lib, variant):
    import torch
    base = util.sample_net()
    cfg = nets.variant(base, spectrogramScaling="db") if variant == "db" else nets.variant(base, windowOverlap=256 - 176)
    dev = torch.device("cuda", 0)
    S_short, S_long = 1 << 20, (1 << 29) + 4096                  # the long row is past 2 GiB
    with SyllableDetector(cfg, channels=1, device=0) as det:
        assert det.geometry.engine == _abi.ENGINE_FUSED          # AUTO commits these to the fold kernel at create time
        det.profile(True)
        x = synth.channels_on_device(1, S_short, dev, fs=cfg.samplingRate)
        det.run(x)
        torch.cuda.synchronize()
        assert [nm for nm, _ in det.lastTimings()][0] == "fused_s_kernel"
        del x
        x = synth.channels_on_device(1, S_long, dev, fs=cfg.samplingRate)
        out, fl = det.run(x)
        torch.cuda.synchronize()
        names = [nm for nm, _ in det.lastTimings()]
        assert not any(nm.startswith("fused") and "spectrogram" not in nm for nm in names), names   # (the DFT half as an STFT is fine for linear columns)
        if variant == "db":
            assert not any(nm.startswith("fused") for nm in names), names
        v = spotcheck.check(det, cfg, x, out, fl, [0], width=96)
        assert v["max_error"] <= 1e-5
