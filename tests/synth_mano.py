"""Re-export: the synthetic MANO-format asset lives in vtaco_amd/synth_mano.py (bench.py's training workload needs it too)."""
from vtaco_amd.synth_mano import PARENTS, as_model, make_asset, write_pkl  # noqa: F401
