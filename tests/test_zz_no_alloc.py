"""
GPU test of psh_rollout_static, the entry point that allocates nothing: its scratch (work-queue counter, packed FNN
weights, QBallBalancer plate angles) lives in a static device array of the library.  It runs the same kernels as
psh_rollout, so the results must be bit-identical, also on a second, smaller call that finds the previous call's
contents in the array.
"""
import numpy as np
import pytest
import torch

from simurlacra_amd import _abi as abi
from tests import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from simurlacra_amd.engine import get_engine
    return get_engine(0)


def _both(eng, case, T, n, **kw):
    """ the first n envs of `case` through psh_rollout and through psh_rollout_static, results preallocated """
    init, dp = eng.f64(case['init'][:, :n]), eng.f64(case['dp'][:, :n])
    args = (case['kind'], case['dt'], T, case['task_c'], case['ps'], eng.f64(case['params']), init, dp)
    ref = eng.rollout(*args, record=True, **kw)
    # rows of a trajectory past its end stay as the caller left them (NaN in `ref`)
    out = {k: torch.full_like(v, float('nan') if v.is_floating_point() else -7) for k, v in ref.items()}
    torch.cuda.synchronize()
    free0, allocs0 = torch.cuda.mem_get_info()[0], eng.alloc_count()
    eng.rollout(*args, record=True, out=out, static_scratch=True, **kw)
    torch.cuda.synchronize()
    assert eng.alloc_count() == allocs0, 'the library allocated scratch during the call'
    assert torch.cuda.mem_get_info()[0] == free0, 'device memory was allocated or freed during the call'
    for k in ref:
        a, b = out[k].cpu().numpy(), ref[k].cpu().numpy()
        np.testing.assert_array_equal(a, b, err_msg=k)   # NaN == NaN here (untouched trajectory rows)
    return ref


@pytest.mark.parametrize('policy,kw', [('fnn', dict(hidden=(32, 32))), ('fnn', dict(hidden=(64, 64, 64, 64), scale=0.15)),
                                       ('linear', {})], ids=['fnn32', 'fnn4x64', 'linear'])
def test_static_rollout_matches_rollout_twice(eng, policy, kw):
    """ QBallBalancer (plate angles from the IK prologue + weight image + queue counter in the static array): a call
    with 3000 envs, then one with 700 envs, each equal to psh_rollout bit for bit """
    kind = abi.ENV_QBB
    case = cases.make_case(kind, 3000, seed=5, policy=policy, **kw)
    r1 = _both(eng, case, 40, 3000)
    r2 = _both(eng, case, 40, 700)
    assert int(r1['length'].sum()) > 0 and int(r2['length'].sum()) > 0
    np.testing.assert_array_equal(r1['ret'][:700].cpu().numpy(), r2['ret'].cpu().numpy())


def test_static_rollout_other_kinds_and_wrappers(eng):
    for kind in (abi.ENV_QQ, abi.ENV_OMO):
        case = cases.make_case(kind, 5000, seed=9, policy='fnn', hidden=(64, 64))
        _both(eng, case, 30, 5000)
        _both(eng, case, 30, 100, wrap=cases.WrapSpec(ops=['act_norm'], noise_seed=3))


def test_static_rollout_rejects_what_does_not_fit(eng):
    from simurlacra_amd.engine import EngineError
    kind = abi.ENV_QBB
    # an ActDelayWrapper needs ring buffers sized by the launch
    case = cases.make_case(kind, 64, seed=1, policy='linear')
    with pytest.raises(EngineError, match='ActDelay'):
        eng.rollout(kind, case['dt'], 5, case['task_c'], case['ps'], case['params'], case['init'], case['dp'],
                    wrap=cases.WrapSpec(ops=['act_delay'], act_delay=2), static_scratch=True)
    # a population of 64x64 networks beyond the weight region
    pop = 8
    case = cases.make_case(kind, pop*4, seed=2, policy='fnn', hidden=(64, 64))
    params = np.tile(case['params'], (pop, 1))
    with pytest.raises(EngineError, match='WEIGHT'):
        eng.rollout(kind, case['dt'], 5, case['task_c'], case['ps'], params, case['init'], case['dp'], policy_group=4,
                    static_scratch=True)
    # more QBallBalancer environments than the plate-angle region holds
    n = abi.PSH_STATIC_MAX_QBB_ENVS + 1
    case = cases.make_case(kind, 8, seed=3, policy='linear')
    init = eng.f64(np.tile(case['init'][:, :1], (1, n)))
    with pytest.raises(EngineError, match='QBB_ENVS'):
        eng.rollout(kind, case['dt'], 2, case['task_c'], case['ps'], case['params'], init, case['dp'][:, :1],
                    dp_index=torch.zeros(n, dtype=torch.int32, device=eng.device), static_scratch=True)
    # the existing entry point still takes all of these
    out = eng.rollout(kind, case['dt'], 2, case['task_c'], case['ps'], case['params'], init, case['dp'][:, :1],
                      dp_index=torch.zeros(n, dtype=torch.int32, device=eng.device))
    assert int(out['length'].min()) == 2


def test_static_rollout_on_a_fresh_context_allocates_nothing(eng):
    """ A context that has never run psh_rollout owns no scratch buffers at all: the static call must get by without them
    (the library counts its own hipMalloc calls per context), with the results of psh_rollout on another context. """
    from simurlacra_amd.engine import Engine
    kind = abi.ENV_QBB
    case = cases.make_case(kind, 2000, seed=11, policy='fnn', hidden=(64, 64))
    fresh = Engine(0)
    try:
        assert fresh.alloc_count() == 0
        init, dp, params = fresh.f64(case['init']), fresh.f64(case['dp']), fresh.f64(case['params'])
        out = fresh.rollout(kind, case['dt'], 30, case['task_c'], case['ps'], params, init, dp, static_scratch=True)
        torch.cuda.synchronize()
        assert fresh.alloc_count() == 0
        ref = eng.rollout(kind, case['dt'], 30, case['task_c'], case['ps'], params, init, dp)
        for k in ('ret', 'length', 'status', 'final_state'):
            np.testing.assert_array_equal(out[k].cpu().numpy(), ref[k].cpu().numpy(), err_msg=k)
        assert np.all(out['length'].cpu().numpy() > 0)
        # the growing entry point does allocate on that context (weight image, plate angles)
        fresh.rollout(kind, case['dt'], 30, case['task_c'], case['ps'], params, init, dp)
        assert fresh.alloc_count() >= 2
    finally:
        torch.cuda.synchronize()
        fresh.close()


def test_static_rollout_captured_into_a_graph_and_replayed(eng):
    """ On a capturing stream the call enqueues its memset and kernels only; the graph is replayed on new inputs written
    into the captured buffers and gives the results of psh_rollout on those inputs. """
    kind, n, T = abi.ENV_QQ, 4096, 50
    case = cases.make_case(kind, 2*n, seed=13, policy='fnn', hidden=(64, 64))
    params = eng.f64(case['params'])
    init, dp = eng.f64(case['init'][:, :n]), eng.f64(case['dp'][:, :n])
    args = (kind, case['dt'], T, case['task_c'], case['ps'], params)
    out = eng.rollout(*args, init, dp, static_scratch=True)   # warm-up outside the capture (module load, attributes)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=eng.device)
    allocs0 = eng.alloc_count()
    with torch.cuda.graph(graph, stream=side):
        eng.rollout(*args, init, dp, out=out, static_scratch=True)
    assert eng.alloc_count() == allocs0
    for half in (0, 1):
        init.copy_(eng.f64(case['init'][:, half*n:(half + 1)*n]))
        dp.copy_(eng.f64(case['dp'][:, half*n:(half + 1)*n]))
        for v in out.values():
            v.fill_(-3)
        graph.replay()
        torch.cuda.synchronize()
        got = {k: v.cpu().numpy().copy() for k, v in out.items()}
        ref = eng.rollout(*args, init, dp)
        for k in got:
            np.testing.assert_array_equal(got[k], ref[k].cpu().numpy(), err_msg=f'{k} (replay {half})')
    assert eng.rollout_ms(0) > 0.   # the eager launches are timed; the captured one recorded no events
