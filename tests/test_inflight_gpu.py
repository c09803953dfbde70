"""pipeline.BatchesInFlight: independent batches on their own streams, no edge between them.  Bar: whatever runs beside a
batch, its vertices and planes are bit-identical to DecodeRenderPlan.step() on the same parameters (which
tests/test_pipeline_gpu.py holds to the CPU oracle on all 64 full-size faces)."""
import pytest
import torch

from conftest import pkg
from gpu_util import net_mod

pytestmark = pytest.mark.gpu
NAMES = ("depth", "texture_image", "normal", "tri_ind")


def _params(synth, B, seeds, dev):
    return [torch.as_tensor(synth.sample_params_batch(B, im_size=200, beta=0.7, seed=s), device=dev) for s in seeds]


@pytest.mark.parametrize("slots", [1, 2, 3])
def test_batches_in_flight_bit_identical_to_the_serial_plan(full_assets, synth, slots):
    dev = torch.device("cuda:0")
    B = 64
    net = net_mod().FaceRecNet(mesh_data=full_assets, batch_size=B, im_size=200, device=dev)
    pipe = pkg("pipeline")
    serial = pipe.DecodeRenderPlan(net, B, 200, 200)
    P = _params(synth, B, (3456, 21, 22, 23, 24, 25, 26), dev)
    want, wantv = [], []
    for p in P:
        want.append([t.clone() for t in serial.step(p)])
        wantv.append(serial.vertex_proj.clone())
    torch.cuda.synchronize()
    fl = pipe.BatchesInFlight(net, B, 200, 200, slots=slots)
    # a stream of seven batches; a slot's results are collected just before the slot comes round again
    pending = {}
    got = {}
    for k, p in enumerate(P):
        i = k % slots
        if i in pending:
            kk, sl = pending.pop(i)
            got[kk] = ([t.clone() for t in sl.wait()], sl.vertex_proj.clone())
        pending[i] = (k, fl.submit(p))
    for i, (kk, sl) in pending.items():
        got[kk] = ([t.clone() for t in sl.wait()], sl.vertex_proj.clone())
    assert sorted(got) == list(range(len(P)))
    for k in range(len(P)):
        assert torch.equal(got[k][1], wantv[k]), "batch %d: vertices differ" % k
        for g, w, n in zip(got[k][0], want[k], NAMES):
            assert torch.equal(g, w), "batch %d, %s: %d elements differ" % (k, n, int((g != w).sum()))
    assert float((want[0][3] >= 0).float().mean()) > 0.2


def test_resident_parameters_many_steps_and_a_consumer_on_the_current_stream(full_assets, synth):
    """The bench's use: parameters resident in each slot, submit() without arguments, sixty batches deep; then a consumer on
    torch's current stream ordered behind a slot with make_current_stream_wait()."""
    dev = torch.device("cuda:0")
    B = 64
    net = net_mod().FaceRecNet(mesh_data=full_assets, batch_size=B, im_size=200, device=dev)
    pipe = pkg("pipeline")
    serial = pipe.DecodeRenderPlan(net, B, 200, 200)
    P = _params(synth, B, (3456, 3457), dev)
    want = [[t.clone() for t in serial.step(p)] for p in P]
    fl = pipe.BatchesInFlight(net, B, 200, 200)
    for sl, p in zip(fl.slots, P):
        sl.params.copy_(p)
    torch.cuda.synchronize()
    for _ in range(60):
        fl.submit()
    sums = []
    for sl in fl.slots:
        sl.make_current_stream_wait()
        sums.append(sl.depth.clamp_min(0).sum())          # runs on the current stream, behind the slot's last launch
    fl.synchronize()
    for i, sl in enumerate(fl.slots):
        for g, w, n in zip(sl.outputs(), want[i], NAMES):
            assert torch.equal(g, w), "slot %d, %s" % (i, n)
        assert torch.equal(sums[i], want[i][0].clamp_min(0).sum())


def test_a_bound_plan_refuses_capture(synth):
    dev = torch.device("cuda:0")
    A = synth.make_assets(grid_u=12, grid_v=14, n_shape=5, n_exp=3, patch=None)
    net = net_mod().FaceRecNet(mesh_data=A, batch_size=2, im_size=24, device=dev)
    fl = pkg("pipeline").BatchesInFlight(net, 2, 24, 24)
    with pytest.raises(RuntimeError):
        fl.slots[0].capture()
    with pytest.raises(ValueError):
        pkg("pipeline").BatchesInFlight(net, 2, 24, 24, slots=0)
