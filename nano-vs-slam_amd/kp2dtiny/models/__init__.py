from .kp2dtiny import (KP2DTINY_CONFIGS, KP2DTINYV3_CONFIGS, KP2DTinyV2, KP2DTinyV3, get_config,  # noqa: F401
                       tiny_factory)
