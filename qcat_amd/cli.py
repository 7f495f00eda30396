"""Command-line driver around the MI355X scanners (the callers and data
formats either side of the hot path).

Same options, batching and output formats as the reference driver (``qcat/cli.py``): FASTA/FASTQ in
(``iter_fastx :235-306``, header split ``:200-214``), batches of 4000 reads with a per-batch kit
vote (``:500-513``), optional trimming (``:521-526``), the minimum-length filter (``:528-530``),
TSV rows (``:408-442``, header ``:486-487``), per-barcode FASTA/FASTQ files or one annotated stream
(``:309-358``) and the end-of-run histogram (``:386-405``).  Written from the behaviour, not the
text, of that file; all alignment work happens in one native call per batch.

    python -m qcat_amd.cli -f reads.fastq -b out_dir --trim
    python -m qcat_amd.cli -f reads.fastq --tsv -k PBC096 > calls.tsv
"""
from __future__ import print_function

import argparse
import io
import itertools
import logging
import os
import sys
import time

from . import __version__, config
from .scanner import factory, get_kits_info, get_modes

BATCH_SIZE = 4000


def get_mode(args):
    if args.MODE_GUPPY:
        logging.warning("Guppy/pyguppy is not available; using the epi2me scan on the GPU.")
        return "epi2me"
    if args.MODE_DUAL:
        return "dual"
    if args.MODE_SIMPLE:
        return "simple"
    return "epi2me"


def parse_args(argv):
    p = argparse.ArgumentParser(prog="qcat-mi355x",
                                description="Demultiplex Oxford Nanopore reads from FASTQ files on an MI355X.")
    p.add_argument("-V", "--version", action="version", version="%(prog)s " + __version__)
    p.add_argument("-l", "--log", dest="log", type=str, default="INFO", help="Print debug information")
    p.add_argument("--quiet", dest="QUIET", action="store_true", help="Don't print summary")
    g = p.add_argument_group("General settings")
    g.add_argument("-f", "--fastq", type=str, dest="fastq", help="Barcoded read file")
    g.add_argument("-b", "--barcode_dir", dest="barcode_dir", type=str, default=None,
                   help="If specified, qcat will demultiplex reads to this folder")
    g.add_argument("-o", "--output", dest="output", type=str, default=None,
                   help="Output file trimmed reads will be written to (default: stdout).")
    g.add_argument("--min-score", dest="min_qual", type=check_minqual_arg, default=None,
                   help="Minimum barcode score. Barcode calls with a lower score will be discarded.")
    g.add_argument("--detect-middle", dest="DETECT_MIDDLE", action="store_true",
                   help="Search for adapters in the whole read")
    g.add_argument("-t", "--threads", dest="threads", type=int, default=1, help="Ignored (the scan runs on the GPU)")
    g.add_argument("--min-read-length", dest="min_length", type=int, default=100,
                   help="Reads short than <min-read-length> after trimming will be discarded.")
    g.add_argument("--tsv", dest="tsv", action="store_true", help="Prints a tsv file containing barcode information "
                                                                 "each read to stdout.")
    g.add_argument("--trim", dest="TRIM", action="store_true", help="Remove adapter and barcode sequences from reads.")
    g.add_argument("-k", "--kit", dest="kit", type=str, default="auto",
                   help="Sequencing kit. Specifying the correct kit will improve sensitivity and specificity "
                        "and runtime (default: auto)")
    g.add_argument("--list-kits", dest="list_kits", action="store_true", help="List all supported kits")
    m = p.add_argument_group("Demultiplexing modes").add_mutually_exclusive_group()
    m.add_argument("--guppy", dest="MODE_GUPPY", action="store_true")
    m.add_argument("--epi2me", dest="MODE_EPI2ME", action="store_true", help="Use EPI2ME's demultiplexing algorithm (default)")
    m.add_argument("--dual", dest="MODE_DUAL", action="store_true", help="Use dual barcoding algorithm")
    m.add_argument("--simple", dest="MODE_SIMPLE", action="store_true",
                   help="Use simple demultiplexing algorithm. Only looks for barcodes, not for adapter sequences. "
                        "Use only for testing purposes!")
    e = p.add_argument_group("EPI2ME options (only valid with --epi2me)")
    e.add_argument("--no-batch", dest="nobatch", action="store_true", help="Don't use information from multiple reads for kit detection")
    e.add_argument("--filter-barcodes", dest="FILTER_BARCODES", action="store_true",
                   help="Filter rare barcode calls when run in batch mode")
    sg = p.add_argument_group("Simple options (only valid with --simple)")
    sg.add_argument("--simple-barcodes", dest="SIMPLE_BARCODES", default="standard",
                    help="Use 12 (standard) or 96 (extended) barcodes for demultiplexing")
    p.add_argument("--device", dest="device", type=int, default=0, help="GPU index")
    return p.parse_args(argv)


def check_minqual_arg(x):
    if x is None:
        return x
    x = float(x)
    if x < 0.0 or x > 100.0:
        raise argparse.ArgumentTypeError("Minimum quality must be between 0 and 100.")
    return x


def split_header(header):
    """FASTA/Q title -> (name, comment or None); tabs count as blanks."""
    cols = header.replace("\t", " ").split(" ")
    return cols[0], (" ".join(cols[1:]) if len(cols) > 1 else None)


def is_fastq(filename):
    if not filename:
        return True                       # stdin is assumed to be FASTQ
    with open(filename) as fh:
        c = fh.read(1)
    if c == "@":
        return True
    if c == ">":
        return False
    raise ValueError("Invalid input file. File must start with '@' or '>'. Current file starts with: " + c)


def _fastq_records_general(lines):
    """(title, seq, qual) the way Biopython's ``FastqGeneralIterator`` (the reference's parser,
    ``qcat/cli.py:260``) reads them: blank lines before a record are skipped, the title is stripped of
    trailing whitespace, sequence and quality may be wrapped over several lines (quality lines are
    collected until they are as long as the sequence, so a quality line may start with '@')."""
    line = next(lines, "")
    while line:
        if line.strip() == "":                                  # blank line between / after records
            line = next(lines, "")
            continue
        if line[0] != "@":
            raise ValueError("Records in Fastq files should start with '@' character")
        title = line[1:].rstrip()
        seq_parts = []
        line = next(lines, "")
        while line and line[0] != "+":
            seq_parts.append(line.rstrip())
            line = next(lines, "")
        if not line:
            raise ValueError("End of file without quality information.")
        second = line[1:].rstrip()
        if second and second != title:
            raise ValueError("Sequence and quality captions differ.")
        seq = "".join(seq_parts)
        if " " in seq or "\t" in seq:                          # (FastqGeneralIterator rejects blanks and tabs)
            raise ValueError("Whitespace is not allowed in the sequence.")
        qual = next(lines, "").rstrip()
        line = next(lines, "")
        while line and len(qual) < len(seq):                   # wrapped quality
            qual += line.rstrip()
            line = next(lines, "")
        if len(qual) != len(seq):
            raise ValueError("Lengths of sequence and quality values differs for %s (%i and %i)." % (title, len(seq), len(qual)))
        yield title, seq, qual


def _fastq_records(handle):
    """FASTQ records -> (title, seq, qual).  Fast path: four lines per step (ONT FASTQ); the first record
    that is not a plain four-line record hands the rest of the file to the general parser above."""
    lines = iter(handle)
    for head, seq, plus, qual in itertools.zip_longest(lines, lines, lines, lines, fillvalue=""):
        seq_s, qual_s = seq.rstrip(), qual.rstrip()
        if head[:1] == "@" and plus[:1] == "+" and len(qual_s) == len(seq_s) and seq_s and " " not in seq_s and "\t" not in seq_s \
                and (len(plus) <= 2 or plus[1:].rstrip() in ("", head[1:].rstrip())):
            yield head[1:].rstrip(), seq_s, qual_s
            continue
        for rec in _fastq_records_general(itertools.chain([head, seq, plus, qual], lines)):
            yield rec
        return


def _fasta_records(handle):
    """(title, seq) like Biopython's ``SimpleFastaParser`` (``qcat/cli.py:287``): text before the first '>'
    is ignored, titles and sequence lines lose trailing whitespace, blanks and carriage returns inside
    the sequence are dropped."""
    title, chunks = None, []
    for line in handle:
        if line.startswith(">"):
            if title is not None:
                yield title, "".join(chunks).replace(" ", "").replace("\r", "")
            title, chunks = line[1:].rstrip(), []
        elif title is not None:
            chunks.append(line.rstrip())
    if title is not None:
        yield title, "".join(chunks).replace(" ", "").replace("\r", "")


class _ChainedRaw(io.RawIOBase):
    """the bytes of several binary streams one after the other (what the native loop read from stdin and did not handle, then
    stdin itself)"""

    def __init__(self, parts):
        io.RawIOBase.__init__(self)
        self.parts = list(parts)

    def readable(self):
        return True

    def readinto(self, b):
        while self.parts:
            data = self.parts[0].read(len(b))
            if data:
                b[:len(data)] = data
                return len(data)
            self.parts.pop(0)
        return 0


def iter_fastx(reads_fx, fastq, batchsize, offset=0, handle=None):
    """Yield (names, comments, seqs, quals) lists of at most ``batchsize`` reads; ``offset``: the byte of the file to start
    at (a record start: where the native loop handed the file back, ``_native_demux``); ``handle``: a text stream to read
    instead (stdin behind what the native loop gave back)."""
    names, comments, seqs, quals = [], [], [], []
    if handle is not None:
        pass
    elif reads_fx and offset:
        # a BYTE offset: seek the binary file, then wrap it -- a text handle only defines seek() for cookies of its own tell()
        # (an arbitrary offset works by accident while the decoder is stateless)
        import io
        raw = open(reads_fx, "rb")
        raw.seek(offset)
        handle = io.TextIOWrapper(raw)
    else:
        handle = open(reads_fx) if reads_fx else sys.stdin
    try:
        records = _fastq_records(handle) if fastq else ((t, s, None) for t, s in _fasta_records(handle))
        try:
            for title, seq, qual in records:
                name, comment = split_header(title)
                names.append(name)
                comments.append(comment)
                seqs.append(seq)
                quals.append(qual)
                if len(names) >= batchsize:
                    yield names, comments, seqs, quals
                    names, comments, seqs, quals = [], [], [], []
        except ValueError as e:
            # a malformed record: the reference logs the parser's message and leaves with status 1 (qcat/cli.py:275-277)
            logging.error(str(e))
            sys.exit(1)
    finally:
        if reads_fx and handle is not sys.stdin:
            handle.close()
    if names:
        yield names, comments, seqs, quals


class _Outputs(object):
    """Per-barcode files (``-b``) or one annotated stream (``-o`` / stdout)."""

    def __init__(self, out_folder, stream, fastq, append=False):
        self.folder, self.stream, self.fastq, self.files = out_folder, stream, fastq, {}
        self.append = append                       # the native loop wrote the first part of the per-barcode files
        if out_folder and not os.path.exists(out_folder):
            os.makedirs(out_folder)

    def write(self, name, comment, sequence, quality, result):
        comment = comment or ""
        quality = quality or ""
        if self.folder:
            key = "none"
            if result["barcode"]:
                key = result["barcode"].name
                if key:
                    key = key.replace("/", "_")
            fh = self.files.get(key)
            if fh is None:
                fh = self.files[key] = open(os.path.join(self.folder, key + (".fastq" if self.fastq else ".fasta")), "a" if self.append else "w")
        else:
            fh = self.stream
            comment = "{} barcode={}".format(comment, str(result["barcode"].id) if result["barcode"] else "none")
        if self.fastq:
            fh.write("@" + name + " " + comment + "\n" + sequence + "\n+\n" + quality + "\n")
        else:
            fh.write(">" + name + " " + comment + "\n" + sequence + "\n")

    def close(self):
        for fh in self.files.values():
            fh.close()


def tsv_row(result, comment, name, sequence):
    if result["barcode"]:
        kit_name = result["adapter"].kit if result["adapter"] else None
        cols = (name, len(sequence), result["barcode"].id, result["barcode_score"], kit_name, result["adapter_end"], comment)
    else:
        cols = (name, len(sequence), "none", "-1", "none", "-1", comment)
    return "\t".join(str(c) for c in cols)


def histogram_lines(barcode_dist, adapter_dist, total_reads):
    lines = ["Adapters detected in %d of %d reads" % (sum(v for k, v in adapter_dist.items() if k != "none"), total_reads)]
    for key in sorted(adapter_dist):
        perc = adapter_dist[key] * 100.0 / total_reads
        lines.append("%15s %6d: | %20s | %6s %%" % (key, adapter_dist[key], int(perc / 5) * "#", "{:.2f}".format(perc)))
    lines.append("Barcodes detected in %d of %d adapters" % (sum(v for k, v in barcode_dist.items() if k != "none"), total_reads))
    for key in sorted(barcode_dist):
        perc = barcode_dist[key] * 100.0 / total_reads
        lines.append("%15s %6d: | %20s | %6s %%" % (key, barcode_dist[key], int(perc / 5) * "#", "{:.2f}".format(perc)))
    return lines


class _FdSink(object):
    """A descriptor native code can write to for a Python text stream: the stream's own descriptor when it has one
    (flushed first), else a temporary file whose content is copied into the stream afterwards."""

    def __init__(self, stream):
        self.stream, self.tmp = stream, None
        try:
            stream.flush()
            self.fd = stream.fileno()
        except (AttributeError, OSError, ValueError):
            import tempfile
            self.tmp = tempfile.TemporaryFile()
            self.fd = self.tmp.fileno()

    def finish(self):
        if self.tmp is not None:
            self.tmp.seek(0)
            while True:
                chunk = self.tmp.read(1 << 24)
                if not chunk:
                    break
                self.stream.write(chunk.decode("ascii"))
            self.tmp.close()


def _native_demux(detector, reads_fq, nobatch, out, tsv, stream, trim, min_read_length, qcat_config, tsv_stream,
                  filter_barcodes=False):
    """The read loop of ``qcat_cli`` inside the native library (``qcat_fastq_demux_stream``, include/qcat_hip.h): the file in
    segments through read | scan | write, whatever its size.  Returns (barcode_dist, adapter_dist, total, skipped,
    resume offset or None) or None when the file or the kit is outside what the native path covers (then nothing has been
    written).  A resume offset means the native loop ended in front of a record that is not a plain one (a wrapped or blank
    line ...): the reads before it are done and written, the caller's own parser takes the rest of the file."""
    from . import native
    layouts = detector.layouts
    if not layouts:
        return None
    one_kit = len(set(l.kit for l in layouts)) == 1
    kit_auto = (not nobatch) and not one_kit            # per-batch vote (detect_barcode_batch); one kit: nothing to vote on
    kit = detector._native_kit(layouts, qcat_config, native.ENDS_BOTH)
    if out and not os.path.exists(out):
        os.makedirs(out)
    tsv_sink = _FdSink(tsv_stream) if tsv else None
    out_sink = _FdSink(stream) if (not out and not tsv) else None
    dual = detector._native_mode == "dual"
    # no file name: the driver's stdin (`cat *.fastq | qcat -b out`, README.md:104 of the reference) -- the descriptor itself; what
    # the native loop reads from a pipe and does not handle comes back in `rest`, in front of the rest of the stream
    stdin_fd, rest = None, None
    if not reads_fq:
        import tempfile
        try:
            stdin_fd = sys.stdin.buffer.fileno()
        except (AttributeError, ValueError, io.UnsupportedOperation):
            return None                                  # (a replaced sys.stdin without a descriptor: the Python loop reads it)
        rest = tempfile.TemporaryFile()
    try:
        bc, ad, n_none, n_ad_none, stats = native.FastqFile.demux_stream(
            reads_fq if reads_fq else None, detector._context(), kit, layouts, dual, batch_size=BATCH_SIZE, kit_auto=kit_auto, trim=trim,
            min_read_length=min_read_length, tsv_fd=tsv_sink.fd if tsv_sink else None, out_fd=out_sink.fd if out_sink else None,
            out_dir=out if out else None, filter_barcodes=bool(filter_barcodes) and not nobatch,
            segment_bytes=int(os.environ.get("QCAT_AMD_SEGMENT_BYTES", "0") or 0),
            input_fd=stdin_fd, rest_fd=rest.fileno() if rest is not None else None)
    except native.FastqFile.Unsupported:
        return None
    for sink in (tsv_sink, out_sink):
        if sink:
            sink.finish()
    # the histograms of qcat/cli.py:366-383 from the native counts (reads the minimum-length filter dropped are not counted)
    adapter_dist, barcode_dist = {}, {}
    for t, cnt in enumerate(ad.tolist()):
        if cnt:
            adapter_dist[layouts[t].kit] = adapter_dist.get(layouts[t].kit, 0) + cnt
    if n_ad_none:
        adapter_dist["none"] = n_ad_none
    if n_none:
        barcode_dist["none"] = n_none
    import numpy as np
    for t, i, j in zip(*np.nonzero(bc)):
        first = layouts[t].get_barcode_set(0)[i]
        if dual:
            second = layouts[t].get_barcode_set(1)[j]
            name = "barcode{:02d}/{:02d}".format(first.id, second.id)
        else:
            name = first.name
        barcode_dist[name] = barcode_dist.get(name, 0) + int(bc[t, i, j])
    resume = stats["next_offset"] if stats["incomplete"] else None
    if rest is not None:
        if stats["incomplete"]:
            import stat
            if stat.S_ISREG(os.fstat(stdin_fd).st_mode):
                os.lseek(stdin_fd, stats["next_offset"], os.SEEK_SET)     # `qcat < file`: read by offset, the descriptor has not moved
            rest.seek(0)
            resume = io.TextIOWrapper(_ChainedRaw([rest, sys.stdin.buffer]))      # (a handle instead of an offset)
        else:
            rest.close()
    return barcode_dist, adapter_dist, stats["n_reads"], stats["n_skipped"], resume


def qcat_cli(reads_fq, kit, mode, nobatch, out, min_qual, tsv, output, threads, trim, adapter_yaml, quiet,
             filter_barcodes, middle_adapter, min_read_length, qcat_config, device=0, tsv_stream=None):
    """Demultiplex one FASTA/FASTQ file; returns (barcode_dist, adapter_dist, total, skipped)."""
    tsv_stream = tsv_stream or sys.stdout
    detector = factory(mode=mode, kit=kit, min_quality=min_qual, kit_folder=adapter_yaml,
                       enable_filter_barcodes=filter_barcodes, scan_middle_adapter=middle_adapter,
                       threads=threads, device=device)
    if tsv:
        print("name", "length", "barcode", "score", "kit", "adapter_end", "comment", sep="\t", file=tsv_stream)
    fastq = is_fastq(reads_fq)
    stream = open(output, "w") if output else sys.stdout
    native_done = None
    if mode in ("epi2me", "dual") and not os.environ.get("QCAT_AMD_NO_NATIVE_FASTQ"):
        # plain four-line FASTQ files and plain two-line FASTA files go through the native ingest / egress
        # (qcat_fastq_demux_stream): same outputs, no Python string per read, --detect-middle and --filter-barcodes included;
        # anything else (stdin, wrapped or odd records, simple mode) stays on -- or comes back to -- the loop below
        native_done = _native_demux(detector, reads_fq, nobatch, out, tsv, stream, trim, min_read_length, qcat_config, tsv_stream,
                                    filter_barcodes=filter_barcodes)
    barcode_dist, adapter_dist, total_reads, skipped_reads, resume = {}, {}, 0, 0, 0
    if native_done is not None:
        barcode_dist, adapter_dist, total_reads, skipped_reads, resume = native_done
    if native_done is not None and resume is None:
        if not quiet:
            for line in histogram_lines(barcode_dist, adapter_dist, total_reads):
                logging.info(line)
            if skipped_reads > 0:
                logging.info("{} reads were skipped due to the min. length filter.".format(skipped_reads))
        if output:
            stream.close()
        return barcode_dist, adapter_dist, total_reads, skipped_reads
    resume_handle = resume if (resume is not None and not isinstance(resume, int)) else None
    outputs = _Outputs(out, stream, fastq, append=(resume is not None and native_done is not None and total_reads > 0))
    for names, comments, seqs, quals in iter_fastx(reads_fq, fastq, 1 if nobatch else BATCH_SIZE, offset=0 if resume_handle else (resume or 0),
                                                   handle=resume_handle):
        if nobatch:
            results = [detector.detect_barcode(read_sequence=seqs[0], read_qualities=quals[0], qcat_config=qcat_config)]
        else:
            results = detector.detect_barcode_batch(read_sequences=seqs, read_qualities=quals, qcat_config=qcat_config)
        for name, comment, sequence, quality, result in zip(names, comments, seqs, quals, results):
            total_reads += 1
            if trim:
                sequence = sequence[result["trim5p"]:result["trim3p"]]
                if quality:
                    quality = quality[result["trim5p"]:result["trim3p"]]
            if len(sequence) < min_read_length:
                skipped_reads += 1
                continue
            bkey = result["barcode"].name if result["barcode"] else "none"
            akey = result["adapter"].kit if result["adapter"] else "none"
            barcode_dist[bkey] = barcode_dist.get(bkey, 0) + 1
            adapter_dist[akey] = adapter_dist.get(akey, 0) + 1
            if tsv:
                print(tsv_row(result, comment, name, sequence), file=tsv_stream)
            if out or not tsv:
                outputs.write(name, comment, sequence, quality, result)
    outputs.close()
    if not quiet:
        for line in histogram_lines(barcode_dist, adapter_dist, total_reads):
            logging.info(line)
        if skipped_reads > 0:
            logging.info("{} reads were skipped due to the min. length filter.".format(skipped_reads))
    if output:
        stream.close()
    return barcode_dist, adapter_dist, total_reads, skipped_reads


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    try:
        args = parse_args(argv)
        level = getattr(logging, args.log.upper(), None)
        if not isinstance(level, int):
            raise ValueError("Invalid log level: %s" % args.log.upper())
        logging.basicConfig(level=level, format="%(message)s")
        if args.list_kits:
            kits = get_kits_info()
            for kit in sorted(kits):
                if kit != "auto" and kit != "DUAL":
                    logging.info("{:<30}{}".format(kit, kits[kit]))
            return
        start = time.time()
        mode = get_mode(args)
        qcat_cli(reads_fq=args.fastq, kit=args.SIMPLE_BARCODES if mode == "simple" else args.kit, mode=mode, nobatch=args.nobatch, out=args.barcode_dir,
                 min_qual=args.min_qual, tsv=args.tsv, output=args.output, threads=args.threads, trim=args.TRIM,
                 adapter_yaml=None, quiet=args.QUIET, filter_barcodes=args.FILTER_BARCODES,
                 middle_adapter=args.DETECT_MIDDLE, min_read_length=args.min_length,
                 qcat_config=config.get_default_config(), device=args.device)
        if not args.QUIET:
            logging.info("Demultiplexing finished in {0:.2f}s".format(time.time() - start))
    except IOError as e:
        logging.error(e)
    except ValueError as e:
        logging.error(e)


if __name__ == "__main__":
    main()
