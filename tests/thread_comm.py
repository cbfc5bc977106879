"""In-process stand-in for a torch.distributed group (moved to miosqp_amd.dist; kept here for the tests' imports)."""
from miosqp_amd.dist import ThreadComm, ThreadWorld  # noqa: F401
