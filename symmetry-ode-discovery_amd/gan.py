"""Import-compatibility alias: the reference keeps these classes in ``gan.py``."""
from .lie import Discriminator, IntParameter, LieGenerator  # noqa: F401
