#!/usr/bin/env python3
"""Generate tests/golden/cli_golden.json: outputs of the UNMODIFIED reference driver
(`qcat.cli.qcat_cli`, /root/reference, read-only) over the four shipped FASTQ files, for the
host-pipeline parity tests.  Dev tool for the authoring container; see
make_golden.py for what is real (all of qcat's Python) and what is a stand-in (parasail -> the
independent scalar DP tests/golden/sg_independent.py, which shares no code with the oracle; Bio -> two parsers)."""
import contextlib
import hashlib
import io
import json
import logging
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402


def sha(path):
    with open(path, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()


def fasta_of_fastq(path):
    """the plain two-line FASTA of a four-line FASTQ file: '>' + title, sequence"""
    with open(path) as fh:
        lines = fh.read().split("\n")
    return "".join(">" + lines[i][1:] + "\n" + lines[i + 1] + "\n" for i in range(0, len(lines) - 3, 4))


def main():
    make_golden.install_standins()
    make_golden.import_reference()
    import qcat.cli as ref_cli
    import qcat.config as ref_config

    class Capture(logging.Handler):
        def __init__(self):
            logging.Handler.__init__(self)
            self.lines = []

        def emit(self, record):
            self.lines.append(record.getMessage())

    cap = Capture()
    logging.getLogger().addHandler(cap)
    logging.getLogger().setLevel(logging.INFO)
    cfg = ref_config.get_default_config()
    runs = []
    data = os.path.join(HERE, "data")
    variants = [
        {"tag": "tsv-auto-batch", "kit": "auto", "mode": "epi2me", "nobatch": False, "tsv": True, "trim": False, "min_len": 100, "dir": False},
        {"tag": "tsv-auto-nobatch-trim", "kit": "auto", "mode": "epi2me", "nobatch": True, "tsv": True, "trim": True, "min_len": 100, "dir": False},
        {"tag": "dir-auto-trim", "kit": "auto", "mode": "epi2me", "nobatch": False, "tsv": False, "trim": True, "min_len": 100, "dir": True},
        {"tag": "stream-kit-trim-minlen1000", "kit": None, "mode": "epi2me", "nobatch": False, "tsv": False, "trim": True, "min_len": 1000, "dir": False},
        {"tag": "tsv-dual", "kit": "auto", "mode": "dual", "nobatch": False, "tsv": True, "trim": False, "min_len": 100, "dir": False},
    ]
    file_kits = {"nbd103.fastq": "NBD104/NBD114", "pbk004.fastq": "PBK004/LWB001",
                 "rab204.fastq": "RAB204/RAB214", "rbk004.fastq": "RBK004"}
    # BASELINE config 1 (ii): the README's 193-read LWB001 example as a generated file (tests/synth.py:
    # config1_fastq; not committed -- the test regenerates it from the seed), through the README's own command shapes
    only_config1 = "--only-config1" in sys.argv
    sys.path.insert(0, os.path.dirname(HERE))
    import synth  # noqa: E402
    import qcat.adapters as ref_adapters
    lwb = [l for l in ref_adapters.populate_adapter_layouts() if l.kit == synth.CONFIG1["kit"]]   # loader order = sorted file names (3p, 5p)
    tmpd = tempfile.mkdtemp(prefix="qcat_cfg1_")
    cfg1 = os.path.join(tmpd, "config1_lwb001_193.fastq")
    with open(cfg1, "w") as fh:
        fh.write(synth.config1_fastq(lwb))
    file_kits_all = dict(file_kits)
    jobs = [] if only_config1 else [(os.path.join(data, f), f, v) for f in sorted(file_kits) for v in variants]
    jobs += [(cfg1, "config1_lwb001_193.fastq", v) for v in variants if v["tag"] in ("tsv-auto-batch", "dir-auto-trim")]
    jobs.append((cfg1, "config1_lwb001_193.fastq",
                 {"tag": "dir-auto-readme", "kit": "auto", "mode": "epi2me", "nobatch": False, "tsv": False, "trim": False, "min_len": 100, "dir": True}))
    file_kits_all["config1_lwb001_193.fastq"] = synth.CONFIG1["kit"]
    # round 5: the driver's --detect-middle / --filter-barcodes (qcat/cli.py:165-171 -> scanner_base.py:593-595, :690-712) on a
    # generated file on which they change the outcome (tests/synth.py: flags_fastq; not committed)
    flags_lays = [l for l in ref_adapters.populate_adapter_layouts() if l.kit == synth.FLAGS["kit"]]
    flags_fq = os.path.join(tmpd, "flags_nbd104_150.fastq")
    with open(flags_fq, "w") as fh:
        fh.write(synth.flags_fastq(flags_lays))
    file_kits_all["flags_nbd104_150.fastq"] = synth.FLAGS["kit"]
    flag_variants = [
        {"tag": "tsv-kit-middle", "kit": None, "mode": "epi2me", "nobatch": False, "tsv": True, "trim": False, "min_len": 100, "dir": False, "middle": True, "filter": False},
        {"tag": "tsv-auto-filter-trim", "kit": "auto", "mode": "epi2me", "nobatch": False, "tsv": True, "trim": True, "min_len": 100, "dir": False, "middle": False, "filter": True},
        {"tag": "dir-auto-middle-filter-trim", "kit": "auto", "mode": "epi2me", "nobatch": False, "tsv": False, "trim": True, "min_len": 0, "dir": True, "middle": True, "filter": True},
        {"tag": "tsv-auto-nobatch-middle-filter", "kit": "auto", "mode": "epi2me", "nobatch": True, "tsv": True, "trim": False, "min_len": 100, "dir": False, "middle": True, "filter": True},
    ]
    # FASTA input (qcat/cli.py:235-306 reads it with SimpleFastaParser, the writers then produce FASTA): the plain two-line
    # FASTA of a shipped FASTQ file, derived here and in the test by fasta_of_fastq (not committed)
    only_fasta = "--only-fasta" in sys.argv
    fa = os.path.join(tmpd, "fasta_of_nbd103.fasta")
    with open(fa, "w") as fh:
        fh.write(fasta_of_fastq(os.path.join(data, "nbd103.fastq")))
    if only_fasta:
        jobs = []
    jobs += [(fa, "fasta_of_nbd103.fasta", v) for v in variants if v["tag"] in ("tsv-auto-batch", "dir-auto-trim", "stream-kit-trim-minlen1000")]
    file_kits_all["fasta_of_nbd103.fasta"] = file_kits["nbd103.fastq"]
    only_flags = "--only-flags" in sys.argv
    if only_flags:
        jobs = []
    if not only_config1 and not only_fasta:
        jobs += [(flags_fq, "flags_nbd104_150.fastq", v) for v in flag_variants]
    if only_fasta:
        with open(os.path.join(HERE, "cli_golden.json")) as fh:
            runs = [r for r in json.load(fh)["runs"] if not r["file"].startswith("fasta_of_")]
    if only_flags:
        with open(os.path.join(HERE, "cli_golden.json")) as fh:
            runs = [r for r in json.load(fh)["runs"] if not r["file"].startswith("flags_")]
    if only_config1:
        with open(os.path.join(HERE, "cli_golden.json")) as fh:
            runs = [r for r in json.load(fh)["runs"] if not r["file"].startswith("config1_")]
    for path, fname, v in jobs:
        if True:                                           # (indentation kept from the per-file loop)
            kit = v["kit"] if v["kit"] else file_kits_all[fname]
            tmp = tempfile.mkdtemp(prefix="qcat_cli_")
            outdir = os.path.join(tmp, "bc") if v["dir"] else None
            outfile = os.path.join(tmp, "out.fastq")
            cap.lines = []
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                ref_cli.qcat_cli(reads_fq=path, kit=kit, mode=v["mode"], nobatch=v["nobatch"],
                                 out=outdir, min_qual=None, tsv=v["tsv"], output=None if v["dir"] else outfile,
                                 threads=1, trim=v["trim"], adapter_yaml=None, quiet=False, filter_barcodes=v.get("filter", False),
                                 middle_adapter=v.get("middle", False), min_read_length=v["min_len"], qcat_config=cfg)
            files = {}
            if outdir:
                for f in sorted(os.listdir(outdir)):
                    files[f] = sha(os.path.join(outdir, f))
            elif os.path.exists(outfile):
                files["out.fastq"] = sha(outfile)
            runs.append({"file": fname, "variant": v, "kit": kit, "stdout": buf.getvalue(),
                         "log": list(cap.lines), "files": files})
            print(fname, v["tag"], len(buf.getvalue()), "bytes stdout,", len(files), "files")
    with open(os.path.join(HERE, "cli_golden.json"), "w") as fh:
        json.dump({"generator": "tests/golden/make_cli_golden.py", "runs": runs}, fh, separators=(",", ":"))
    print("cli_golden.json", os.path.getsize(os.path.join(HERE, "cli_golden.json")), "bytes")


if __name__ == "__main__":
    main()
