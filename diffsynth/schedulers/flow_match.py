"""`diffsynth.schedulers.flow_match.FlowMatchScheduler` (reference: schedulers/flow_match.py:5-125)."""
from physicedit_amd.scheduler import FlowMatchScheduler  # noqa: F401
