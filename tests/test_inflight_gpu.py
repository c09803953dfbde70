"""pipeline.BatchesInFlight: independent batches on their own streams, no edge between them.  Bar: whatever runs beside a
batch, its vertices and planes are bit-identical to DecodeRenderPlan.step() on the same parameters (which
tests/test_pipeline_gpu.py holds to the CPU oracle on all 64 full-size faces)."""
import numpy as np
import pytest
import torch

from conftest import pkg
from gpu_util import assert_render_equal, net_mod

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
    # (round 6: plans that run beside another batch ask for 8-row strips where the library alone picks 10 -- a scheduling hint,
    # so this comparison with the serial plan's 10-row strips is also the proof that it changes no bit)
    assert fl.strip_rows == (8 if slots > 1 else 0) and all(sl.strip_rows == fl.strip_rows for sl in fl.slots) and serial.strip_rows == 0
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


@pytest.mark.parametrize("B,S,small", [(64, 200, False), (5, 96, True), (3, 37, True)])
def test_strip_height_hint_changes_no_bit(full_assets, small_assets, synth, B, S, small):
    """FR_PHASES_STRIP_ROWS (bits 8-15 of fr_decode_render_forward's `phases`): any strip height the binned rasteriser serves, and
    any it does not (those are ignored), gives the planes of the library's own choice, bit for bit."""
    dev = torch.device("cuda:0")
    net = net_mod().FaceRecNet(mesh_data=small_assets if small else full_assets, batch_size=B, im_size=S, device=dev)
    pipe = pkg("pipeline")
    P = torch.as_tensor(synth.sample_params_batch(B, im_size=S, n_shape=net.ndim_shape, n_exp=net.ndim_exp, beta=0.7, seed=77), device=dev)
    base = pipe.DecodeRenderPlan(net, B, S, S)
    want = [t.clone() for t in base.step(P)]
    assert float((want[3] >= 0).float().mean()) > 0.0005      # (something is on screen)
    for rows in (8, 5, 4, 25, 2, 1, 200, 255):
        plan = pipe.DecodeRenderPlan(net, B, S, S, strip_rows=rows)
        got = plan.step(P)
        torch.cuda.synchronize()
        for g, w, n in zip(got, want, NAMES):
            assert torch.equal(g, w), "strip_rows=%d, %s: %d elements differ" % (rows, n, int((g != w).sum()))
    hint = pipe.BatchesInFlight.strip_rows_in_flight(net, B, S, S)
    assert hint == 8 if (B, S) == (64, 200) else 0 <= hint <= 255


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


@pytest.mark.parametrize("arith", ["f32", "q30l4"])
def test_in_flight_slots_against_the_oracle_directly(oracle, full_assets, synth, arith):
    """the in-flight route against the CPU oracle itself, not only against the serial plan.  Two slots, eight
    full-size faces each, six submits deep so that the batches really run beside each other; then each slot's vertices against
    oracle.decode_3dmm (the slot's arithmetic; in-kernel rotation: <= 2 ulp and >= 99 % bit-equal, the bar of
    tests/test_decode_gpu.py) and its four planes against oracle.render_depth on those very vertices, bit for bit."""
    dev = torch.device("cuda:0")
    B = 8
    h = pkg("_lib")
    prev, prev_lv = h.decode_arith(), h.q30_levels()
    if arith != "f32":
        h.set_decode_arith(h.DECODE_ARITH_Q30, int(arith[-1]))
    try:
        net = net_mod().FaceRecNet(mesh_data=full_assets, batch_size=B, im_size=200, device=dev)
        fl = pkg("pipeline").BatchesInFlight(net, B, 200, 200, slots=2)
        assert fl.slots[0].q30 == (0 if arith == "f32" else int(arith[-1]))
        Ps = [synth.sample_params_batch(B, im_size=200, beta=0.7, seed=s) for s in (3456, 4567)]
        for sl, P in zip(fl.slots, Ps):
            sl.params.copy_(torch.as_tensor(P, device=dev))
        torch.cuda.synchronize()
        for _ in range(6):
            fl.submit()
        fl.synchronize()
    finally:
        h.set_decode_arith(prev, prev_lv)
    A = full_assets
    for sl, P in zip(fl.slots, Ps):
        V = sl.vertex_proj.contiguous().cpu().numpy()
        Vo = oracle.decode_3dmm(P, A["mu"], A["pc_shape"], A["pc_exp"], 200.0, R=oracle.rotation_matrix_batch(P[:, :3]), q30=sl.q30)
        a, o = V.view(np.int32).astype(np.int64), Vo.view(np.int32).astype(np.int64)
        ulp = np.abs(np.where(a < 0, -(a & 0x7FFFFFFF), a) - np.where(o < 0, -(o & 0x7FFFFFFF), o))
        assert ulp.max() <= 2 and (ulp == 0).mean() >= 0.99
        want = oracle.render_depth(V, A["tri"], A["vertex"][None], 200, 200)
        assert_render_equal(tuple(t.cpu().numpy() for t in sl.outputs()), want, "in-flight slot (%s)" % arith)
        assert float((want[3] >= 0).mean()) > 0.2


def test_a_bound_plan_step_orders_its_parameter_copy(full_assets, synth):
    """DecodeRenderPlan.step(params) on a plan BOUND to a stream must copy `params` on that stream, behind
    the current stream that produced them -- the decode otherwise reads the plan's buffer before or while the copy lands."""
    dev = torch.device("cuda:0")
    B = 16
    net = net_mod().FaceRecNet(mesh_data=full_assets, batch_size=B, im_size=200, device=dev)
    pipe = pkg("pipeline")
    serial = pipe.DecodeRenderPlan(net, B, 200, 200)
    st = torch.cuda.Stream(device=dev)
    bound = pipe.DecodeRenderPlan(net, B, 200, 200, stream=st)
    P = [torch.as_tensor(synth.sample_params_batch(B, im_size=200, beta=0.7, seed=s), device=dev) for s in range(40, 46)]
    for p in P:
        want = [t.clone() for t in serial.step(p)]
        # the producer of the parameters runs on the CURRENT stream right before step(): a long-ish elementwise chain
        q = p.clone()
        for _ in range(20):
            q = q * 1.0 + 0.0
        outs = bound.step(q)
        del q                                     # (record_stream keeps its memory until the bound stream's copy has run)
        st.synchronize()
        for g, w, n in zip(outs, want, NAMES):
            assert torch.equal(g, w), n
    del bound                                     # (__del__ drains the bound stream before the buffers are released)


def test_a_bound_plan_refuses_capture(synth):
    dev = torch.device("cuda:0")
    A = synth.make_assets(grid_u=12, grid_v=14, n_shape=5, n_exp=3, patch=None)
    net = net_mod().FaceRecNet(mesh_data=A, batch_size=2, im_size=24, device=dev)
    fl = pkg("pipeline").BatchesInFlight(net, 2, 24, 24)
    with pytest.raises(RuntimeError):
        fl.slots[0].capture()
    with pytest.raises(ValueError):
        pkg("pipeline").BatchesInFlight(net, 2, 24, 24, slots=0)
