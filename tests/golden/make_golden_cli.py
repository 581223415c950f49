#!/usr/bin/env python3
"""Capture the reference CLI's surface as data (build container only): every option string, its
type, default and whether it is a store_true switch, read from the parser that
/root/reference/point_vs/parse_args.py builds; plus the model kwargs point_vs.py:189-221 derives
from two sample command lines. Written to tests/golden/cli_flags.json."""
import argparse
import importlib.util
import json
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
spec = importlib.util.spec_from_file_location('ref_parse_args', '/root/reference/point_vs/parse_args.py')
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

captured = {}
_orig = argparse.ArgumentParser.parse_args


def _capture(self, *a, **k):
    captured['parser'] = self
    return _orig(self, *a, **k)


argparse.ArgumentParser.parse_args = _capture
sys.argv = ['point_vs.py', 'egnn', '/tmp/x']
mod.parse_args()
argparse.ArgumentParser.parse_args = _orig
flags = []
for act in captured['parser']._actions:
    if isinstance(act, argparse._HelpAction):
        continue
    flags.append({'names': list(act.option_strings) or [act.dest], 'dest': act.dest,
                  'type': None if act.type is None else act.type.__name__,
                  'default': act.default, 'store_true': isinstance(act, argparse._StoreTrueAction)})
(HERE / 'cli_flags.json').write_text(json.dumps({'flags': flags}, indent=1) + '\n')
print(len(flags), 'flags')
