"""Stand-in for the reference's `cuda/py_nvcc_utils.py` (/root/reference/src/cuda/py_nvcc_utils.py:7-44) for callers
that swap this package in by putting it first on `sys.path`.

Every reference app does `import cuda.py_nvcc_utils as py_nvcc_utils`, `py_nvcc_utils.add_args(parser)` and
`py_nvcc_utils.config_compiler(args)` in its constructor (`src/run_live_layered.py:10, 23-27`; `src/3d_bz.py`,
`src/run_live.py`, `src/train_model.py` likewise).  There the two calls point PyCUDA's `SourceModule` at a
directory of fatbins (`:7-23`) and `get_module(name)` compiles or loads one `.cu` module (`:25-44`).  Here nothing
is compiled at run time: the kernels are in the prebuilt `csrc/librdf_hip.so`, loaded once by `_lib.load()`.  So

* `add_args` registers the same two options, so that a command line written for the reference still parses;
* `config_compiler` keeps the reference's one check (both options at once is an error, `:14`) and ignores the rest;
* `get_module` has nothing to hand out -- the classes of `decision_tree`, `cuda.points_ops` and `cuda.mean_shift`
  call the C ABI themselves -- and says so loudly instead of returning something that fails later.
"""
from .. import _lib


def add_args(parser):
    """The reference's `--fatbin_in` / `--fatbin_out` (`py_nvcc_utils.py:7-9`): accepted, not used."""
    parser.add_argument('--fatbin_in', nargs='?', required=False, type=str,
                        help='(accepted for compatibility with 3d-beats; the HIP kernels are prebuilt, nothing is read)')
    parser.add_argument('--fatbin_out', nargs='?', required=False, type=str,
                        help='(accepted for compatibility with 3d-beats; the HIP kernels are prebuilt, nothing is written)')


def config_compiler(args):
    """No compiler to configure.  The reference asserts that not both directories are given (`py_nvcc_utils.py:14`);
    so does this, so that a wrong command line fails in the same place."""
    f_in = getattr(args, 'fatbin_in', None)
    f_out = getattr(args, 'fatbin_out', None)
    assert not (f_in and f_out)


def get_module(n):
    """The reference returns a PyCUDA module whose `get_function(name)` gives a launchable kernel
    (`py_nvcc_utils.py:25-37`).  This package has no such object: its kernels are entry points of librdf_hip.so."""
    raise _lib.RdfError(
        f"py_nvcc_utils.get_module({n!r}): there is no run-time compiled module in this package. The kernels of "
        f"'{n}' are entry points of the prebuilt {_lib.library_path()} (include/rdf_hip.h); use "
        "decision_tree.DecisionTreeEvaluator / LayeredDecisionForest, cuda.points_ops.PointsOps, "
        "cuda.mean_shift.MeanShift, or bind the C ABI with `_lib.load()`.")
