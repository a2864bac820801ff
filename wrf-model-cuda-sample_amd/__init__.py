"""MI355X-native advance_mu_t (WRF-ARW acoustic sub-step: mu, ww, theta update).

Host-side mirror of the reference's interface for this one path:

* ``advance_mu_t(...)``  -- same name, argument order and meaning as
  ``SUBROUTINE advance_mu_t`` (module_small_step_em.f90:7-18); numpy arrays take the
  one-shot host drop-in, torch CUDA tensors the device-resident one.  Both go through
  the C-ABI of ``include/amt_advance_mu_t.h`` into the hand-written gfx950 kernels.
* ``GridConfig``         -- the three ``grid_config_rec_type`` logicals the routine reads.
* ``synth``              -- seeded WRF-shaped synthetic inputs (host and device fill).
* ``wrfdump``            -- the reference drivers' big-endian per-variable dump format.
* ``patch``              -- resident device patch (torch-allocated arrays) and j-slab
  decomposition with the one-row input-halo exchange over torch.distributed (RCCL).

The directory name contains '-' (it is the reference repository's name + ``_amd``), so it
is loaded through ``__graft_entry__.load_package()`` under the module name
``wrf_model_cuda_sample_amd``.  There is no CPU fallback: without the built HIP library
every compute entry point raises.
"""
from .config import GridConfig  # noqa: F401
from .lib import AmtError, load_library, library_path  # noqa: F401
from .api import advance_mu_t, bind_device_call, compute_window, VARIANT_AUTO, VARIANT_COLUMN, VARIANT_MARCH, LAUNCH_BESIDE_OTHERS  # noqa: F401
from .api import host_cache_enable, host_invalidate, host_defer, host_fetch, host_stale, host_release, held_arrays  # noqa: F401
from .api import host_set_devices, host_devices  # noqa: F401
from . import synth  # noqa: F401
from . import patch  # noqa: F401
from . import wrfdump  # noqa: F401
