#!/usr/bin/env python
"""Regenerate the golden GAF files from the UNMODIFIED reference built by oracle/Makefile.
Run in the build container (needs oracle/_ref/minigraph):  python tests/golden/make_golden.py"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "..", "oracle", "_ref", "minigraph")

for q in ("MT-orangA.fa", "MT-chimp.fa", "MT-human.fa"):
    out = os.path.join(HERE, q.replace(".fa", ".cx_lr.gaf"))
    with open(out, "wb") as fo:
        subprocess.check_call([REF, "-cx", "lr", os.path.join(HERE, "MT.gfa"), os.path.join(HERE, q)], stdout=fo, stderr=subprocess.DEVNULL)
    print(out, os.path.getsize(out))
