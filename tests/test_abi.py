"""
CPU tests of the drop-in boundary: libpysimhip.so loads, exports every symbol include/pysimhip.h declares, the
ctypes mirror matches the header, and argument errors are reported without touching a GPU.
"""
import ctypes as C
import os
import re

import pytest

from simurlacra_amd import _abi as abi
from simurlacra_amd.engine import load_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header():
    with open(os.path.join(ROOT, 'include', 'pysimhip.h')) as f:
        return f.read()


def test_library_exports_every_declared_symbol():
    lib = load_library()
    declared = set(re.findall(r'\b(psh_[a-z_0-9]+)\s*\(', _header()))
    assert declared == set(abi.PSH_SYMBOLS), declared ^ set(abi.PSH_SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.psh_version() == abi.PSH_VERSION


def test_env_dims_match_the_python_table():
    lib = load_library()
    for kind, dims in abi.ENV_DIMS.items():
        out = (C.c_int32*5)()
        assert lib.psh_env_dims(kind, C.byref(out)) == 0
        assert tuple(out) == dims
    assert lib.psh_env_dims(99, C.byref((C.c_int32*5)())) == abi.PSH_ERR_ARG


def test_constants_of_the_ctypes_mirror_match_the_header():
    h = _header()
    for name in ('PSH_MAX_STATE', 'PSH_MAX_OBS', 'PSH_MAX_ACT', 'PSH_MAX_DP', 'PSH_MAX_HIDDEN', 'PSH_MAX_WIDTH',
                 'PSH_MAX_FEATS', 'PSH_MAX_DISTR', 'PSH_MAX_DELAY'):
        m = re.search(rf'#define {name} (\d+)', h)
        assert m and int(m.group(1)) == getattr(abi, name), name


def test_struct_layouts_match_the_c_compiler(tmp_path):
    """ sizeof/offsetof of every ABI struct as gcc lays them out vs the ctypes mirror """
    import subprocess
    structs = {'psh_task': abi.PshTask, 'psh_policy': abi.PshPolicy, 'psh_wrappers': abi.PshWrappers,
               'psh_rollout_desc': abi.PshRolloutDesc,
               'psh_step_desc': abi.PshStepDesc, 'psh_distr': abi.PshDistr, 'psh_rand_desc': abi.PshRandDesc,
               'psh_init_desc': abi.PshInitDesc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "pysimhip.h"', 'int main(void){']
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['return 0;}']
    src = tmp_path/'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path/'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f'{cname}.{fname}']) == getattr(cls, fname).offset, f'{cname}.{fname}'


@pytest.mark.skipif(os.path.exists('/dev/kfd'), reason='this check is for boxes without a GPU')
def test_create_fails_loudly_without_a_gpu():
    lib = load_library()
    h = C.c_void_p()
    assert lib.psh_create(0, C.byref(h)) == abi.PSH_ERR_NODEV
    from simurlacra_amd.engine import Engine, EngineError
    with pytest.raises(EngineError):
        Engine(0)


def test_philox_streams_of_the_samplers_are_pairwise_disjoint():
    """ Domain randomization, init-state sampling, wrapper noise, exploration noise and the stochastic reset may all be
    keyed by the same (seed, env offset): their stream ids must differ, and stay below the stride (16) by which a
    sampler that draws several pairs per step advances its id. """
    with open(os.path.join(ROOT, 'simurlacra_amd', 'csrc', 'psh_rng.h')) as f:
        ids = dict(re.findall(r'#define (PSH_STREAM_[A-Z_]+) (\d+)u', f.read()))
    assert set(ids) == {'PSH_STREAM_RANDOMIZE', 'PSH_STREAM_OBS_NOISE', 'PSH_STREAM_ACT_NOISE', 'PSH_STREAM_EXPL',
                        'PSH_STREAM_BERNOULLI', 'PSH_STREAM_INIT', 'PSH_STREAM_BOOTSTRAP'}
    vals = [int(v) for v in ids.values()]
    assert len(set(vals)) == len(vals) and max(vals) < 16
    # the oracle draws initial states from the same stream: component r of env e is uniform number 0 of
    # Philox(counter = (e, 0, r, PSH_STREAM_INIT), key = seed)
    import numpy as np
    from oracle import oracle_lib as orc
    from simurlacra_amd import envdefs
    dp = np.array(list(envdefs.nominal_domain_param(abi.ENV_OMO).values()))[:, None].repeat(3, axis=1)
    init = orc.sample_init(abi.ENV_OMO, 3, dp, seed=7, env_offset=40)
    o = orc.philox([41, 0, 0, int(ids['PSH_STREAM_INIT'])], [7, 0])
    u = ((o[0] >> 5)*67108864.0 + (o[1] >> 6))/9007199254740992.0
    assert init[0, 1] == -0.75 + u*(-0.65 - -0.75)
