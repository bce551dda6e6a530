"""Convenience: build a frozen config the way the reference entry points do
(``get_cfg(); get_cfg_defaults(cfg); cfg.merge_from_file(...); cfg.merge_from_list(opts)``,
reference demo/demo.py:124-141)."""
from __future__ import annotations

import os
from typing import Iterable, Optional

from .config import CfgNode, get_cfg, get_cfg_defaults

CONFIG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs")


def make_cfg(config_file: Optional[str] = None, opts: Optional[Iterable] = None, freeze: bool = True) -> CfgNode:
    cfg = get_cfg()
    get_cfg_defaults(cfg)
    if config_file:
        if not os.path.isabs(config_file) and not os.path.exists(config_file):
            config_file = os.path.join(CONFIG_DIR, config_file)
        cfg.merge_from_file(config_file)
    cfg.merge_from_list(list(opts or []))
    if freeze:
        cfg.freeze()
    return cfg
