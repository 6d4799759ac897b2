"""Mirror of the reference's ``mxnext`` helper surface (mxnext/simple.py, mxnext/complicate.py) over rangedet_amd.mx."""
from .simple import *  # noqa: F401,F403
from .complicate import normalizer_factory  # noqa: F401
