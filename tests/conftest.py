import json
import os
import sys

# Before torch (libgomp) loads: idle OpenMP workers SLEEP instead of spinning.  The whole-model CPU tests run 8 torch
# threads; with the default active wait policy any other load on the box (measured: four busy cores next to the suite)
# turns their fork-join barriers into spin contention -- the README-recipe test goes from 18 s to 164 s, MobileBERT from 3 s
# to 139 s, the CPU suite from 3 to 9 minutes.  Passive waiting costs ~10 % on an idle box and keeps the suite at 3-4
# minutes on a busy one.
os.environ.setdefault('OMP_WAIT_POLICY', 'PASSIVE')

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'transformer-quantization_amd')
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
    config.addinivalue_line('markers', "default_route: runs with the product's default options.INT8_LINEAR ('auto')")


@pytest.fixture(autouse=True)
def _pin_the_route_under_test(request):
    """The product default is options.INT8_LINEAR = 'auto': with autograd off, fixed-range forwards take the exact-integer /
    fused route (quantization/options.py).  Almost every parity test here is a statement about ONE route -- the layered
    module chain against the oracle / the reference fixtures, or the integer route (switched on explicitly) against the
    integer oracle -- so each test starts from the layered route (False) and the tests of the DEFAULT are marked
    `default_route`.  The switch is restored after every test whatever it did."""
    from harness import bert, mobilebert
    from quantization import options
    switches = [(bert.QSelfAttention, 'fuse'), (bert.QResidualBlock, 'fuse'), (bert.QLayer, 'fuse_ffn'), (bert.QEmbeddings, 'fuse'),
                (mobilebert.QBottleneckLayer, 'fuse'), (mobilebert.QMobileSelfAttention, 'fuse'),
                (mobilebert.QResidualNoNorm, 'fuse'), (mobilebert.QFFN, 'fuse'), (mobilebert.QMobileLayer, 'fuse_ffn')]
    before = options.INT8_LINEAR
    default = request.node.get_closest_marker('default_route') is not None
    options.INT8_LINEAR = 'auto' if default else False
    for cls, attr in switches:
        setattr(cls, attr, None if default else False)      # the harness models' tri-state switches (None: follow the option)
    yield
    options.INT8_LINEAR = before
    for cls, attr in switches:
        setattr(cls, attr, None)


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


# Files whose tests launch other processes (torchrun workers, bench.py ranks, IPC peers): collected LAST, so that a
# launcher / rendezvous flake under `-x` can never hide a parity test or the CLI harness behind it.
_LAUNCHER_FILES = ('test_dist_gloo.py', 'test_mailbox.py', 'test_rccl_single_rank.py', 'test_rccl_raw.py')


def pytest_collection_modifyitems(config, items):
    items.sort(key=lambda it: os.path.basename(str(it.fspath)) in _LAUNCHER_FILES)      # stable: keeps file order
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    meta = json.loads(str(z['meta'])) if 'meta' in z.files else None
    return z, meta


@pytest.fixture(scope='session')
def golden_fake_quant():
    return load_golden('fake_quant')


@pytest.fixture(scope='session')
def golden_estimators():
    return load_golden('estimators')


@pytest.fixture(scope='session')
def golden_toy():
    return load_golden('toy_model')


@pytest.fixture(scope='session')
def golden_adaround():
    return load_golden('adaround')


def check_weights_reproduced(hf, z):
    """The harness model carries exactly the parameters the fixture was generated on: both sides fill them from numpy's
    legacy Mersenne-Twister stream (harness/weights.py), which does not depend on the torch / transformers build -- so
    this is an ASSERTION (round 3: a different build could only skip here, silently dropping every whole-model parity
    test).  Check sum: float64 sum of |w| over all parameters in name order (numpy pairwise sums: thread-count
    independent), compared to 1e-12 relative."""
    from harness.weights import weight_check_sum
    got, want = weight_check_sum(hf), float(z['weight_check_sum'])
    assert abs(got - want) <= 1e-12 * abs(want), ('harness weights differ from the fixture', got, want, str(z['versions']))
