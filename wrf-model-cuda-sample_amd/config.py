"""The part of WRF's ``grid_config_rec_type`` that advance_mu_t reads.

The reference type has 1 796 fields (module_configure.f90:3-1800); the routine uses three
logicals (module_small_step_em.f90:97-103; module_configure.f90:434,436,447), which is also
all the C version's struct carries (advance_mu_t.h:3-8).
"""
from dataclasses import dataclass


@dataclass(frozen=True)
class GridConfig:
    periodic_x: bool = False
    specified: bool = False
    nested: bool = False

    def as_ints(self):
        """(periodic_x, specified, nested) as 0/1 -- the order of the C-ABI."""
        return int(self.periodic_x), int(self.specified), int(self.nested)


def flags_as_ints(config_flags):
    if isinstance(config_flags, GridConfig):
        return config_flags.as_ints()
    if isinstance(config_flags, (tuple, list)):
        px, sp, ne = config_flags
    elif isinstance(config_flags, dict):
        px, sp, ne = (config_flags.get(k, False) for k in ("periodic_x", "specified", "nested"))
    else:
        px, sp, ne = config_flags.periodic_x, config_flags.specified, config_flags.nested
    return int(bool(px)), int(bool(sp)), int(bool(ne))
