"""TEST INFRASTRUCTURE ONLY.  The seeded FreqCodec checkpoints and recipe configs of the FreqCodec golden cases live with the
product's other synthetic-checkpoint helpers (funcodec_amd/synth.py, funcodec_amd/config.py: bench.py needs them too and may
not import oracle/); this module keeps the names the oracle-side scripts use."""
from funcodec_amd.config import freq_recipe_config  # noqa: F401
from funcodec_amd.synth import freq_plan, make_freq_state_dict  # noqa: F401
