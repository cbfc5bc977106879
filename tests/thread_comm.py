"""The in-process communicator under the names the CPU tests use (ranks = threads of one process)."""
from miosqp_amd.poolcomm import PoolComm as ThreadComm, PoolWorld as ThreadWorld  # noqa: F401
