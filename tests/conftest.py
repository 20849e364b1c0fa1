import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session', autouse=True)
def _built_library():
    """Tests need the in-tree libpnpmri.so.  The driver builds it first (`__graft_entry__.build()`);
    if a fresh checkout lacks it and hipcc is present, build it here (the product itself never
    builds or falls back: pnp_admm_cnc_mri_amd._lib raises when the library is missing)."""
    import shutil
    lib = os.path.join(ROOT, 'pnp_admm_cnc_mri_amd', 'libpnpmri.so')
    if not os.path.exists(lib) and (shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc')):
        import __graft_entry__
        __graft_entry__.build()
    yield


# `-m gpu` on a box without a GPU must fail loudly, not silently skip: there is deliberately no
# auto-skip of gpu-marked tests here.


@pytest.fixture(scope='session')
def golden_inputs():
    d = np.load(os.path.join(GOLD, 'inputs_set1_05.npz'))
    masks = {k[:-len('_packbits')]: np.unpackbits(d[k])[:65536].reshape(256, 256)
             for k in d.files if k.endswith('_packbits')}
    return {'gray': d['gray_u8'], 'noises': d['noises_c128'] * 3.0, 'masks': masks}


@pytest.fixture(scope='session')
def golden_admm():
    return np.load(os.path.join(GOLD, 'admm_set1_05.npz'))


@pytest.fixture(scope='session')
def known_answers():
    with open(os.path.join(GOLD, 'known_answers.json')) as f:
        return json.load(f)


def rel_l2(a, b):
    """||a - b|| / ||b|| in float64 / complex128 (complex inputs keep their imaginary parts)."""
    a, b = np.asarray(a), np.asarray(b)
    dt = np.complex128 if (np.iscomplexobj(a) or np.iscomplexobj(b)) else np.float64
    a, b = a.astype(dt), b.astype(dt)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def weights50(name):
    """The contractive fixture weights behind tests/golden/pnp50_set1_05.npz (oracle/make_golden_pnp.py --fifty): rebuilt bit for bit from
    the committed seeds and operator-norm gains; IRCNN -> its 25-model bank {str(index): state_dict} (S6:196)."""
    from pnp_admm_cnc_mri_amd import denoisers as D
    meta = json.load(open(os.path.join(GOLD, 'pnp_known.json')))
    seed, gains = meta['known50']['seeds'][name], meta['gains50']
    net, _, _ = D.build(name)
    fam = D.family(name)
    if fam == 'ircnn':
        return {str(i): D.contractive_state_dict(net, fam, seed + i, gains['%s/%d' % (name, i)]) for i in range(25)}
    return D.contractive_state_dict(net, fam, seed, gains[name])


def weights_trained(name='ffdnet_gray'):
    """The TRAINED fixture network (oracle/train_fixture_denoiser.py: FFDNet-gray trained KAIR-style on seeded synthetic images for a few
    minutes; tests/golden/ffdnet_gray_trained.npz) as a state_dict -- the weights behind the 'trained_*' goldens of pnp50_set1_05.npz."""
    import torch
    assert name in ('ffdnet_gray', 'dncnn_25')                     # dncnn_25 (round 6): DnCNN-17, the x - n(x) family, trained at sigma = 25 / 255
    w = np.load(os.path.join(GOLD, name + '_trained.npz'))
    return {k: torch.from_numpy(w[k]) for k in w.files}
