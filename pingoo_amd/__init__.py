"""pingoo_amd — MI355X-native batched WAF rule matching for Pingoo's per-request rule path."""
from . import _abi  # noqa: F401
from .batch import GEOIP_DTYPE, VERDICT_DTYPE, Request, RequestBatch, geoip_entries  # noqa: F401
