#!/usr/bin/env python3
"""Generates tests/golden/abi_layout.json: sizeof / offsetof of every struct of the drop-in boundary, measured by a C probe
compiled against the REFERENCE's own headers (/root/reference/minigraph.h, gfa.h, mgpriv.h).  Run where /root/reference exists;
tests/test_abi.py compiles the same probe against include/minigraph_amd.h and compares (and re-derives this file when the reference is
present).  The probe text lives in tests/abi_probe.py so that both sides compile the same thing."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import abi_probe  # noqa: E402

if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    lay = abi_probe.run_probe(["-I" + ref], ['#include "minigraph.h"', '#include "gfa.h"', '#include "mgpriv.h"'])
    json.dump(lay, open(os.path.join(HERE, "abi_layout.json"), "w"), indent=1, sort_keys=True)
    print("%d entries" % len(lay))
