"""GPU test of utils/graph.py: the train step replayed from a HIP graph does the same work as the eager step.

Two models start from identical weights; one runs three eager steps (forward with batch-stat BN, PSMLoss, backward, SGD), the other
three replays of the captured step (its warm-up runs are undone by restoring the state first).  Losses and updated weights must agree
to the run-to-run noise of the few atomicAdd reductions left in the backward (soft-argmin / classifier-weight adjoints)."""
import copy

import pytest
import torch

from disprcnn_amd.utils import synth

pytestmark = pytest.mark.gpu


def test_graphed_train_step_matches_eager():
    from disprcnn_amd.modeling.psmnet.stackhourglass import PSMNet
    from disprcnn_amd.utils.graph import GraphedStep
    from disprcnn_amd.utils.loss_utils import PSMLoss
    dev = torch.device("cuda:0")
    n = 6
    base = PSMNet(48, 0)
    base.load_state_dict(synth.synth_state_dict(base.state_dict()), strict=True)
    fl, fr = synth.synth_features(n, 32, 28, 28, tag="graphA")
    fl, fr = fl.to(dev), fr.to(dev)
    tgt = synth.hash_uniform("graphA:t", (n, 112, 112), 0.0, 47.0).to(dev)
    msk = torch.ones_like(tgt, dtype=torch.uint8)
    crit = PSMLoss()

    def make():
        m = copy.deepcopy(base).to(dev).train()
        opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9)

        def step():
            opt.zero_grad(set_to_none=True)
            loss = crit(m.forward_from_features(fl, fr, (112, 112)), {"disparity": tgt, "mask": msk})
            loss.backward()
            opt.step()
            return loss
        return m, opt, step

    def snapshot(m):
        return {k: p.detach().clone() for k, p in m.named_parameters()}

    m0, _, step0 = make()
    eager = [step0().item()]
    w_eager = snapshot(m0)                           # after ONE step: the tight comparison (later steps amplify 1-ulp differences)
    eager += [step0().item() for _ in range(2)]
    m2, _, step2 = make()                            # noise floor: a second, independent eager step (a few atomicAdd reductions remain)
    step2()
    w_floor = snapshot(m2)

    m1, opt1, step1 = make()
    state = copy.deepcopy(m1.state_dict())
    gs = GraphedStep(step1, warmup=2)
    m1.load_state_dict(state)                       # undo the warm-up and capture-time updates (in place: same storage)
    for grp in opt1.param_groups:
        for p in grp["params"]:
            st = opt1.state.get(p)
            if st and st.get("momentum_buffer") is not None:
                st["momentum_buffer"].zero_()
    graphed = [gs().item()]
    w_graph = snapshot(m1)
    graphed += [gs().item() for _ in range(2)]

    for a, b in zip(eager, graphed):
        assert abs(a - b) <= 2e-4 * max(abs(a), 1.0), (eager, graphed)
    assert eager[0] != eager[2]                      # the weights did move
    w_base = {k: p.detach().to(dev) for k, p in base.named_parameters()}
    for name, ref in w_eager.items():
        moved = (ref - w_base[name]).abs().max().item()
        d = (w_graph[name] - ref).abs().max().item()
        floor = (w_floor[name] - ref).abs().max().item()
        assert d <= 10 * floor + 1e-4 * moved + 1e-9, (name, d, floor, moved)
    assert int(m1.dres0[0][1].num_batches_tracked) == int(m0.dres0[0][1].num_batches_tracked) == 3   # in-kernel counter replays too
