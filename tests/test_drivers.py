"""train.py / generate_result.py counterparts: flags, schedule and the S3D key remap (CPU)."""
import torch

from vinet_amd import generate_result as GR
from vinet_amd import train as TR


def test_train_flags_match_reference_defaults():
    a = TR.build_parser().parse_args([])
    assert (a.no_epochs, a.lr, a.batch_size, a.clip_size, a.log_interval, a.no_workers) == (40, 1e-4, 8, 32, 5, 4)
    assert a.kldiv is True and a.cc is False and a.sim is False and a.l1 is False
    assert (a.kldiv_coeff, a.cc_coeff, a.sim_coeff) == (1.0, -1.0, -1.0)
    assert a.decoder_upsample == 1 and a.num_hier == 3 and a.load_weight == "None" and a.dataset == "DHF1KDataset"
    assert TR.build_parser().parse_args(["--cc", "True"]).cc is True


def test_sliding_window_schedule_matches_generate_result():
    """generate_result.py:58-73 for T=4, N=9: frames 3..8 from forward clips, frames 0..2 from flipped ones"""
    T, N = 4, 9
    sched = GR.sliding_window_schedule(N, T)
    fwd = [(o, c) for o, c, f in sched if not f]
    rev = [(o, c) for o, c, f in sched if f]
    assert fwd == [(i, list(range(i - T + 1, i + 1))) for i in range(T - 1, N)]
    assert rev == [(i - T + 1, list(range(i, i - T, -1))) for i in range(T - 1, 2 * T - 2)]
    assert sorted(o for o, _, _ in sched) == list(range(N))           # every frame predicted exactly once
    # call order: the flipped call directly follows its forward call (generate_result.py:67-71)
    assert [f for _, _, f in sched[:6]] == [False, True, False, True, False, True]
    assert GR.sliding_window_schedule(2 * T - 2, T) == []               # "more frames are needed"


def test_s3d_kinetics_key_remap():
    from vinet_amd import model
    bb = model.BackBoneS3D()
    src = {}
    want = {"base.0.conv_s.weight": "base1.0.conv_s.weight", "base.3.conv_t.weight": "base1.3.conv_t.weight",
            "base.5.branch0.0.conv.weight": "base2.0.branch0.0.conv.weight",
            "module.base.8.branch1.1.conv_s.weight": "base3.0.branch1.1.conv_s.weight",
            "base.15.branch3.1.bn.running_mean": "base4.1.branch3.1.bn.running_mean"}
    sd = bb.state_dict()
    for i, (k, dst) in enumerate(want.items()):
        src[k] = torch.full_like(sd[dst], float(i + 1))
    TR.remap_s3d_kinetics(src, bb)
    sd = bb.state_dict()
    for i, dst in enumerate(want.values()):
        assert torch.equal(sd[dst], torch.full_like(sd[dst], float(i + 1)))
