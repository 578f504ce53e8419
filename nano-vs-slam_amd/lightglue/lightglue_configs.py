"""Matcher configurations per extractor family (reference: lightglue/lightglue_configs.py:1-29; the values are the
reference's data)."""

_COMMON = {"name": "lightglue", "n_layers": 4}          # "name": just for interfacing
LIGHT_GLUE_CONFIGS = {
    "S": dict(_COMMON, input_dim=32, descriptor_dim=32),
    "F": dict(_COMMON, input_dim=64, descriptor_dim=64),
    "A": dict(_COMMON, input_dim=32, descriptor_dim=32),
}


def get_light_glue_config(config):
    if config not in LIGHT_GLUE_CONFIGS:
        raise ValueError("Config not supported")
    return dict(LIGHT_GLUE_CONFIGS[config])
