"""`UserConfig` with the reference's behaviour (utils/user_config.py:13-25): two YAML files merged, the
model file overriding the data file; a missing key reads as None."""
from collections import UserDict

import yaml


def load_yaml(path):
    with open(path, "r", encoding="utf-8") as f:
        return yaml.load(f, Loader=yaml.FullLoader)


class UserConfig(UserDict):
    def __init__(self, common, model):
        custom = load_yaml(common)
        custom.update(load_yaml(model))
        super().__init__(custom)

    def __missing__(self, key):
        return None
