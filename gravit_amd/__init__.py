"""gravit_amd -- MI355X (gfx950) engine adapter for GraviT: gvt::render::adapter::hip.

Only the adapter hot path lives here: the C-ABI library (csrc/, include/gvt_hip.h),
the host-side mirror of GraviT's Adapter interface (adapter.py) and of its Image/Domain schedulers
(scheduler.py), and the scene inputs (scenes.py).  Nothing in this package imports the CPU oracle.
"""
from . import layouts  # noqa: F401

__all__ = ["layouts", "scenes", "capi", "adapter", "scheduler"]
