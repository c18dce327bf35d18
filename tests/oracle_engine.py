"""The CPU oracle behind GnnEngine's interface lives in oracle/engine.py (shared with bench.py's cpu_baseline leg)."""
from oracle.engine import OracleEngine  # noqa: F401
