"""Import-name alias: ``import carmcmc as cm`` works as with the reference package
(src/carmcmc/__init__.py re-exports ``_carmcmc.*`` and the Python API); everything comes from
``carma_pack_amd``."""
from carma_pack_amd._carmcmc import *  # noqa: F401,F403
from carma_pack_amd.carma_pack import *  # noqa: F401,F403
from carma_pack_amd import _carmcmc  # noqa: F401
