#!/usr/bin/env python3
"""haslr.py — whole-pipeline driver: long-read preparation, short-read assembly, overlap trimming, alignment, haslr_assemble.

The command line, the file names inside the output directory, the progress text on stdout, the exit codes and the command line of
every tool are those of the reference's driver (bin/haslr.py:18-50 the five steps, :54-158 and :199-262 the tool invocations,
:287-378 the options), so an output directory started by either driver is finished by the other: every step is skipped when its
product is already there. Installed by the build as haslr_amd/bin/haslr.py next to haslr_assemble (the MI355X assembler) and
minia_nooverlap; Minia, minimap2 and fastutils are external programs and are looked up in that directory first (where the reference
expects all five, bin/haslr.py:24-34) and then on PATH.

Two deliberate differences: a step's captured stdout goes to "<product>.part" and is renamed when the tool succeeds, so a failed or
interrupted tool does not leave a truncated product that the next run would take for finished; and HASLR_ASSEMBLE_ARGS (split like a
shell would) is appended to haslr_assemble's command line, e.g. "--device 1".

Pinned against the reference driver run over recording stand-in tools (tests/test_driver.py, tests/golden/driver/).
"""
import argparse
import datetime
import glob
import multiprocessing
import os
import shlex
import shutil
import subprocess
import sys

VERSION = '0.8a1'
MINIMAP2_PRESET = {'corrected': '-k19', 'pacbio': '-Hk17', 'nanopore': '-k15'}     # bin/haslr.py:90-96
TOOLS = ('haslr_assemble', 'minia_nooverlap', 'fastutils', 'minia', 'minimap2')    # checked in this order (bin/haslr.py:24-34)
OWN_TOOLS = ('haslr_assemble', 'minia_nooverlap')


def say(text):
    sys.stdout.write(text)
    sys.stdout.flush()


def stamp():
    return datetime.datetime.now().strftime('%d-%b-%Y %H:%M:%S')


def fail(code, text):
    say(text)
    sys.exit(code)


class Layout:
    """every path the pipeline touches, from the options (the names are the contract between runs and between drivers)"""

    def __init__(self, a):
        out = a.out
        self.lr_fofn = out + '/lr.fofn'
        self.sr_fofn = out + '/sr.fofn'
        self.lr_name = 'lrall' if a.cov_lr == 0 else 'lr{0}x'.format(a.cov_lr)
        self.lr_file = '{0}/{1}.fasta'.format(out, self.lr_name)
        self.sr_prefix = out + '/sr_k{}_a{}'.format(a.minia_kmer, a.minia_solid)
        self.sr_asm = self.sr_prefix + '.' + a.minia_asm + '.fa'
        self.sr_noov = self.sr_prefix + '.' + a.minia_asm + '.nooverlap.fa'
        self.sr_good = '{0}.{1}.nooverlap.{2}.fa'.format(self.sr_prefix, a.minia_asm, a.min_src)
        tag = '{0}_k{1}_a{2}_c{3}_{4}'.format(a.minia_asm, a.minia_kmer, a.minia_solid, a.min_src, self.lr_name)
        self.map_base = '{0}/map_{1}'.format(out, tag)
        self.asm_dir = '{0}/asm_{1}_b{2}_s{3}_sim{4}'.format(out, tag, a.aln_block, a.edge_sup, a.aln_sim)


class Step:
    """one tool run: announce, skip when the product exists, otherwise run and report like the reference does"""

    def __init__(self, title, product, tool, argv, stdout=None, stderr=None):
        self.title, self.product, self.tool, self.argv = title, product, tool, argv
        self.stdout, self.stderr = stdout, stderr          # a path, 'product' (= capture into the product), subprocess.DEVNULL or None

    def run(self, paths):
        say('[{0}] {1}... '.format(stamp(), self.title))
        if os.path.isfile(self.product):
            say('already exists\n')
            return
        opened = []

        def sink(spec):
            if spec is None or spec == subprocess.DEVNULL:
                return spec
            path = self.product + '.part' if spec == 'product' else spec
            for f in opened:
                if f.name == path:
                    return f
            opened.append(open(path, 'w'))
            return opened[-1]

        try:
            done = subprocess.run([paths[self.tool]] + self.argv, stdout=sink(self.stdout), stderr=sink(self.stderr))
        except OSError as err:
            fail(os.EX_SOFTWARE, 'failed\nERROR: {}\n'.format(err))
        finally:
            for f in opened:
                f.close()
        if done.returncode != 0:
            fail(os.EX_SOFTWARE, 'failed\nERROR: "{}" returned non-zero exit status\n'.format(self.tool))
        if self.stdout == 'product':
            os.replace(self.product + '.part', self.product)
        say('done\n')


def write_list(path, names, names_are_lists):
    """a file of file names: the names themselves, or the lines of the given lists (--short-fofn / --long-fofn)"""
    with open(path, 'w') as fp:
        for fn in names:
            if names_are_lists:
                with open(fn, 'r') as src:
                    for line in src:
                        fp.write(line)
            else:
                fp.write(fn + '\n')


def plan(a, lay):
    """the steps in order, each a callable"""
    t = str(a.threads)
    steps = []

    # 1. long reads: numeric names, optionally the longest cov_lr x genome bases (bin/haslr.py:199-258)
    def long_reads():
        if not os.path.isfile(lay.lr_fofn):
            write_list(lay.lr_fofn, a.long, a.long_fofn)
        if a.cov_lr == 0:
            return Step('renaming long reads and storing in {0}'.format(lay.lr_file), lay.lr_file, 'fastutils',
                        ['format', '-i', lay.lr_fofn, '-d', '--fofn'], stdout='product')
        return Step('subsampling {0}x long reads to {1}'.format(a.cov_lr, lay.lr_file), lay.lr_file, 'fastutils',
                    ['subsample', '-i', lay.lr_fofn, '-d', str(a.cov_lr), '-g', a.genome, '-lnk', '--fofn'], stdout='product')
    steps.append(long_reads)

    # 2. short-read assembly, unless contigs were given (bin/haslr.py:160-195)
    if a.contig is None:
        def short_reads():
            if not os.path.isfile(lay.sr_asm):
                write_list(lay.sr_fofn, a.short, a.short_fofn)
            log = lay.sr_prefix + '.log'
            return Step('assembling short reads using Minia', lay.sr_asm, 'minia',
                        ['-nb-cores', t, '-out-dir', a.out, '-out-tmp', a.out, '-out', lay.sr_prefix, '-in', lay.sr_fofn,
                         '-kmer-size', str(a.minia_kmer), '-abundance-min', str(a.minia_solid), '-no-ec-removal'], stdout=log, stderr=log)
        steps.append(short_reads)

        def drop_glue():
            for fn in glob.glob(lay.sr_prefix + '.unitigs.fa.glue*'):
                try:
                    os.remove(fn)
                except OSError:
                    say('ERROR: cannot delete file: {}\n'.format(fn))
        steps.append(drop_glue)

    # 3. overlap trimming, then the length filter (bin/haslr.py:115-156)
    source = a.contig if a.contig is not None else lay.sr_asm
    steps.append(lambda: Step('removing overlaps in short read assembly', lay.sr_noov, 'minia_nooverlap',
                              [source, str(a.minia_kmer)], stdout='product', stderr=subprocess.DEVNULL))
    steps.append(lambda: Step('removing short sequences in short read assembly', lay.sr_good, 'fastutils',
                              ['format', '-i', lay.sr_noov, '-m', str(a.min_src), '-c'], stdout='product', stderr=subprocess.DEVNULL))

    # 4. long reads against the kept contigs (bin/haslr.py:82-111)
    steps.append(lambda: Step('aligning long reads to short read assembly using minimap2', lay.map_base + '.paf', 'minimap2',
                              ['-t', t, '--secondary=no', '-c', MINIMAP2_PRESET[a.type], lay.sr_good, lay.lr_file],
                              stdout='product', stderr=lay.map_base + '.log'))

    # 5. the assembler: done when asm.final.fa is there (bin/haslr.py:54-78)
    extra = shlex.split(os.environ.get('HASLR_ASSEMBLE_ARGS', ''))
    steps.append(lambda: Step('assembling long reads using HASLR', lay.asm_dir + '/asm.final.fa', 'haslr_assemble',
                              ['-t', t, '-c', lay.sr_noov, '-l', lay.lr_file, '-m', lay.map_base + '.paf', '-d', lay.asm_dir,
                               '--aln-block', str(a.aln_block), '--aln-sim', str(a.aln_sim), '--edge-sup', str(a.edge_sup)] + extra,
                              stdout=lay.asm_dir + '.out', stderr=lay.asm_dir + '.err'))
    return steps


def locate_tools(here):
    """{tool: path}; each tool must answer `-h` with exit status 0 (bin/haslr.py:262-278)"""
    paths = {}
    for tool in TOOLS:
        prog = os.path.join(here, tool)
        if not os.path.isfile(prog) and tool not in OWN_TOOLS:
            prog = shutil.which(tool) or prog
        say('checking {}: '.format(prog))
        if not os.path.isfile(prog):
            fail(os.EX_SOFTWARE, 'not found\n')
        try:
            rc = subprocess.run([prog, '-h'], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode
        except OSError as err:
            fail(os.EX_SOFTWARE, 'failed\nERROR: {}\n'.format(err))
        if rc != 0:
            fail(os.EX_SOFTWARE, 'failed\n')
        say('ok\n')
        paths[tool] = prog
    return paths


class HelpLayout(argparse.HelpFormatter):
    """one 'flags METAVAR' column instead of argparse's repeated metavars"""

    def __init__(self, prog):
        super().__init__(prog, max_help_position=50, width=1000)

    def _format_action_invocation(self, action):
        if not action.option_strings or action.nargs == 0:
            return super()._format_action_invocation(action)
        return ', '.join(action.option_strings) + ' ' + self._format_args(action, self._get_default_metavar_for_optional(action))


def options(argv):
    me = os.path.basename(__file__)
    p = argparse.ArgumentParser(add_help=False, formatter_class=HelpLayout,
                                usage='haslr.py [-t THREADS] -o OUT_DIR -g GENOME_SIZE -l LONG [LONG ...] -x LONG_TYPE -s SHORT [SHORT ...]')
    req = p.add_argument_group(title='required arguments')
    req.add_argument('-o', '--out', type=str, required=True, metavar='OUT_DIR', help='output directory')
    req.add_argument('-g', '--genome', type=str, required=True, metavar='GENOME_SIZE', help='estimated genome size; accepted suffixes are k,m,g')
    req.add_argument('-l', '--long', type=str, nargs='+', help='long read file')
    req.add_argument('-x', '--type', type=str, required=True, choices=['pacbio', 'nanopore', 'corrected'], metavar='LONG_TYPE',
                     help='type of long reads chosen from {pacbio, nanopore, corrected}')
    req.add_argument('-s', '--short', type=str, nargs='+', help='short read file. Required if --contig is not given')
    req.add_argument('-c', '--contig', type=str, help='pre-assembled short read contigs. If given, no need to pass --short')
    opt = p.add_argument_group(title='optional arguments')
    opt.add_argument('-t', '--threads', type=int, default=1, help='number of CPU threads to use [1]')
    opt.add_argument('--cov-lr', type=int, default=25, help='amount of long read coverage to use for assembly (0 for using all long reads) [25]')
    opt.add_argument('--aln-block', type=int, default=500, help='minimum length of alignment block [500]')
    opt.add_argument('--aln-sim', type=float, default=0.85, help='minimum alignment similarity [0.85]')
    opt.add_argument('--edge-sup', type=int, default=3, help='minimum number of long read supporting each edge [3]')
    opt.add_argument('--minia-kmer', type=int, default=49, help='kmer size used by minia [49]')
    opt.add_argument('--minia-solid', type=int, default=3, help='minimum kmer abundance used by minia [3]')
    opt.add_argument('--minia-asm', type=str, default='contigs', choices=['contigs', 'unitigs'], metavar='MINIA_ASM',
                     help='type of minia assembly chosen from {contigs,unitigs} [contigs]')
    opt.add_argument('--min-src', type=int, default=250, help='minimum length of short read contigs to be used [250]')
    opt.add_argument('--short-fofn', default=False, action='store_true', help='SHORT is a file of file names')
    opt.add_argument('--long-fofn', default=False, action='store_true', help='LONG is a file of file names')
    opt.add_argument('-v', '--version', action='version', version=VERSION, help='print version')
    opt.add_argument('-h', '--help', action='help', help='show this help message and exit')
    if not argv:
        p.print_usage()
        sys.exit(os.EX_USAGE)
    a = p.parse_args(argv)
    if a.long is None:
        fail(os.EX_USAGE, '{0}: error: argument -l/--long is required\n'.format(me))
    if a.short is None and a.contig is None:
        fail(os.EX_USAGE, '{0}: error: either -s/--short or -c/--contig is required for "{1}"\n'.format(me, a.type))
    a.threads = min(max(a.threads, 1), multiprocessing.cpu_count())
    for fn in a.long + (a.short or []) + ([a.contig] if a.contig is not None else []):
        if not os.path.isfile(fn):
            fail(os.EX_USAGE, '{0}: error: could not find file {1}\n'.format(me, fn))
    a.out = os.path.abspath(a.out)
    a.long = [os.path.abspath(f) for f in a.long]
    if a.short is not None:
        a.short = [os.path.abspath(f) for f in a.short]
    if a.contig is not None:
        a.contig = os.path.abspath(a.contig)
    return a


def main(argv=None):
    a = options(sys.argv[1:] if argv is None else argv)
    paths = locate_tools(os.path.dirname(os.path.abspath(__file__)))
    say('number of threads: {}\n'.format(a.threads))
    say('output directory: {}\n'.format(a.out))
    os.makedirs(a.out, exist_ok=True)
    for make in plan(a, Layout(a)):
        step = make()
        if step is not None:
            step.run(paths)
    sys.exit(os.EX_OK)


if __name__ == '__main__':
    main()
