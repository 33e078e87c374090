"""YAML experiment configs with recursive `_base_` includes -> attribute dicts.

Keeps the reference's schema and loader behaviour (utils/config.py:19-46;
schema in SURVEY.md Appendix D) without easydict, which is absent here."""
import os

import yaml


class AttrDict(dict):
    """dict with attribute access; nested dicts are converted on assignment."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, key, value):
        if isinstance(value, dict) and not isinstance(value, AttrDict):
            value = AttrDict(value)
        elif isinstance(value, (list, tuple)):
            value = type(value)(AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v
                                for v in value)
        super().__setitem__(key, value)

    __setattr__ = __setitem__

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key) from None


def _merge(node, new, root_dir):
    for key, val in new.items():
        if key == '_base_' and not isinstance(val, dict):
            path = val if os.path.isabs(val) or os.path.exists(val) else os.path.join(root_dir, val)
            with open(path, 'r') as f:
                val = yaml.safe_load(f)
            node[key] = AttrDict()
            _merge(node[key], val, root_dir)
        elif isinstance(val, dict):
            if key not in node or not isinstance(node[key], dict):
                node[key] = AttrDict()
            _merge(node[key], val, root_dir)
        else:
            node[key] = val
    return node


def cfg_from_yaml_file(cfg_file):
    """utils/config.py:38-46.  `_base_` paths are taken relative to the working
    directory like the reference does, else relative to the repo root."""
    with open(cfg_file, 'r') as f:
        new = yaml.safe_load(f)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return _merge(AttrDict(), new, root)


def get_config(args, logger=None):
    """utils/config.py:48-59: on --resume re-read the yaml saved in the
    experiment directory; otherwise rank 0 saves a copy there."""
    if getattr(args, 'resume', False):
        cfg_path = os.path.join(args.experiment_path, 'config.yaml')
        if not os.path.exists(cfg_path):
            raise FileNotFoundError(cfg_path)
        args.config = cfg_path
    config = cfg_from_yaml_file(args.config)
    if not getattr(args, 'resume', False) and getattr(args, 'local_rank', 0) == 0 \
            and getattr(args, 'experiment_path', None):
        os.makedirs(args.experiment_path, exist_ok=True)
        import shutil
        shutil.copyfile(args.config, os.path.join(args.experiment_path, 'config.yaml'))
    return config
