"""Command line front end mirroring the reference's `chord-detect` (chord_detect.py:11-67):
same flags, same output lines.  `--displayplots` is accepted and ignored (no matplotlib here).

Interface mirror of sevagh/chord-detection's chord_detection/chord_detect.py (MIT License,
Copyright (c) Sevag Hanssian): flag names, help strings and the print format are the reference's on purpose;
`--device` and `--note-names` are additions."""
import argparse
import sys

from .multipitch import METHODS
from . import esacf, harmonic_energy, iterative_f0, prime_multif0  # noqa: F401  (registers the methods)


def main_cli(argv=None):
    method_nums_help_string = "-1 = all, "
    for k in METHODS.keys():
        method_nums_help_string += "{0} ({1}), ".format(k, METHODS[k].display_name())
    method_nums_help_string = method_nums_help_string[:-2]

    parser = argparse.ArgumentParser(
        prog="chord-detection",
        description="Collection of chord-detection techniques",
        formatter_class=argparse.RawDescriptionHelpFormatter,
    )
    parser.add_argument("--key", action="store_true",
                        help="estimate the key using the Krumhansl-Schmuckler key-finding algorithm")
    parser.add_argument("--displayplots", type=int, default=-1,
                        help="accepted for compatibility; intermediate plots are not produced")
    parser.add_argument("--method", type=int, help=method_nums_help_string, default=next(iter(METHODS.keys())))
    parser.add_argument("--device", type=int, default=0, help="GPU index")
    parser.add_argument("--note-names", choices=("unicode", "ascii"), default="unicode",
                        help="how the librosa the reference runs with spells sharps: 'unicode' (librosa >= 0.8; "
                             "methods 1, 3, 4 then drop C#, D#, F#, G#, A#, as the reference does today) or "
                             "'ascii' (librosa < 0.8; every pitch class counts, as in the reference's README)")
    parser.add_argument("input_path", help="Path to WAV audio clip")
    args = parser.parse_args(argv)

    compute_objs = []
    if args.method == -1:
        for v in METHODS.values():
            compute_objs.append(v(args.input_path, device=args.device, note_names=args.note_names))
    else:
        try:
            compute_objs.append(METHODS[args.method](args.input_path, device=args.device,
                                                     note_names=args.note_names))
        except KeyError:
            raise ValueError("valid methods: {0}".format(method_nums_help_string))

    for compute_obj in compute_objs:
        print("{0} - {1}".format(compute_obj.method_number(), compute_obj.display_name()))
        chromagram = compute_obj.compute_pitches(args.displayplots)
        print(chromagram)
        if args.key:
            print(chromagram.key())
    return 0


if __name__ == "__main__":
    sys.exit(main_cli())
