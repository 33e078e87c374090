"""Name -> class registry with `build(cfg)` dispatch on cfg.NAME
(utils/registry.py:246-288, models/build.py:4-15 of the reference)."""


class Registry:
    def __init__(self, name):
        self._name = name
        self._classes = {}

    @property
    def name(self):
        return self._name

    def __contains__(self, key):
        return key in self._classes

    def get(self, key):
        return self._classes.get(key)

    def register_module(self, name=None, force=False, module=None):
        def _do(cls):
            key = name or cls.__name__
            if not force and key in self._classes:
                raise KeyError(f'{key} is already registered in {self._name}')
            self._classes[key] = cls
            return cls
        if module is not None:
            return _do(module)
        return _do

    def build(self, cfg, **kwargs):
        if not isinstance(cfg, dict) or 'NAME' not in cfg:
            raise KeyError(f'`cfg` must be a dict holding the key "NAME", got {cfg}')
        cls = self.get(cfg['NAME'])
        if cls is None:
            raise KeyError(f"{cfg['NAME']} is not in the {self._name} registry")
        return cls(cfg, **kwargs)


MODELS = Registry('models')
DATASETS = Registry('dataset')


def build_model_from_cfg(cfg, **kwargs):
    return MODELS.build(cfg, **kwargs)
