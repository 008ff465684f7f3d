"""Import-level drop-in (SURVEY 8b Face 1): the reference's callers import ``models.*``, ``utils.*`` and ``data.*.config_*`` as
TOP-LEVEL packages.  The import blocks of the three entry points are executed verbatim against the repo root, in a fresh
interpreter (so the top-level names cannot leak into the test session).  Third-party lines are left out: ``munch`` (absent here;
``load_config`` returns an attribute dict with the same access) and ``pyro`` -- ``pyro.infer.SVI / Trace_ELBO`` and
``pyro.optim.Adam`` are replaced by ``structured_latent_odes_amd.svi.{SVI, Trace_ELBO, Adam}`` by design."""
import ast
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

# the first-party import lines of the reference entry points (training_cvs.py:8-15, training_proc.py:12-18, training_challenge.py:12-20)
BLOCKS = {
    "training_cvs.py": [
        "from data.cvs.config_cvs import load_config",
        "from utils.ODE_dataset import create_transforms, ODEDataCSV",
        "from utils.utils import set_seed",
        "from utils.plotting import individual_cvs, visualize_latent",
        "from models.mechanistic_cvs import MechanisticModel",
        "from models.mechanistic_cvs_Gauss import MechanisticModelGauss",
    ],
    "training_proc.py": [
        "from utils.utils import set_seed",
        "from data.proc.config_proc import load_config",
        "from models.mechanistic_proc import MechanisticModel",
        "from models.mechanistic_proc_Gauss import MechanisticModelGauss",
        "from utils.proc_dataset import build_datasets",
        "from utils.plotting import individual_proc, visualize_latent",
    ],
    "training_challenge.py": [
        "from utils.ODE_dataset import create_transforms, ODEDataChallenge",
        "from utils.utils import set_seed",
        "from data.challenge.config_challenge import load_config",
        "from models.mechanistic_challenge import MechanisticModel",
        "from models.mechanistic_challenge_Gauss import MechanisticModelGauss",
        "from data.challenge.challenge_data import build_datasets",
        "from utils.plotting import individual_challenge, visualize_latent",
    ],
}
INNER = [  # imports between the reference's own modules that a caller may also reach for
    "from models.blackbox_ode import OdeModel, OdeFunc, Dynamics",
    "from models.encoder_conv import EncoderCONV, Exp",
    "from models.encoder_mlp import EncoderMLP, ListOutModule, ConcatModule, call_nn_op",
    "from models.decoders import Decoder, GaussianDecoder, VarianceGaussianDecoder",
    "from utils.exp import Exp",
    "from utils.utils import find_norm_params",
    "from utils.proc_dataset import depth",
    "from data.proc.config_proc import Config",
    "from data.proc.load_proc_data import load",
]
CHECK = """
import structured_latent_odes_amd.models._mechanistic as M
assert issubclass(MechanisticModel, M.MechanisticBase) and issubclass(MechanisticModelGauss, MechanisticModel)
cfg = load_config()
assert cfg.solver == "midpoint" and cfg.adjoint_solver is True and cfg["ode_hidden_dim"] == 25 and cfg.data_path
set_seed(cfg.seed)
print("ok")
"""


def _run(lines, check=""):
    code = "import sys\nassert sys.path[0] == %r\n%s\n%s" % (ROOT, "\n".join(lines), check)
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n" % ROOT + code], cwd="/tmp", env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


@pytest.mark.parametrize("script", sorted(BLOCKS))
def test_reference_import_block_resolves_at_top_level(script):
    assert "ok" in _run(BLOCKS[script], CHECK)


def test_inner_module_imports_resolve():
    _run(INNER)


@pytest.mark.parametrize("script", sorted(BLOCKS))
def test_import_block_is_the_reference_s(script):
    """The lists above are the reference's own first-party import lines (checked where the reference is present)."""
    path = os.path.join(REF, script)
    if not os.path.exists(path):
        pytest.skip("reference tree not present on this box")
    want = []
    for node in ast.parse(open(path).read()).body:
        if isinstance(node, ast.ImportFrom) and node.module.split(".")[0] in ("models", "utils", "data"):
            want.append("from %s import %s" % (node.module, ", ".join(a.name for a in node.names)))
    assert want == BLOCKS[script]
