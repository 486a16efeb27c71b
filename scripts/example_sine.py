#!/usr/bin/env python3
"""BASELINE config 1 (plumbing): a 1-voice sine -> gain at 48 kHz in 256-frame blocks.

The reference's `scripts/example_sine.py:41-57` plays `amplitude * sin(2*pi*frequency*t)` through
sounddevice (absent here); this is the same signal pulled block by block through the node API
(`Fixed -> Sine -> Gain`) by the headless BlockDriver and checked against that formula.

    python scripts/example_sine.py [FREQUENCY] [-a AMPLITUDE] [-n BLOCKS]
"""
import argparse
import pathlib
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))


def main(argv=None) -> float:
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument('frequency', nargs='?', type=float, default=500, help='frequency in Hz (default: %(default)s)')
    parser.add_argument('-a', '--amplitude', type=float, default=0.2, help='amplitude (default: %(default)s)')
    parser.add_argument('-n', '--blocks', type=int, default=8)
    args = parser.parse_args(argv)

    from signals_amd.chain.driver import BlockDriver
    from signals_amd.chain.fixed import Fixed
    from signals_amd.chain.fx import Gain
    from signals_amd.chain.osc import Sine

    hertz = Fixed()
    hertz.get_state().value = np.array([[args.frequency]])
    amplitude = Fixed()
    amplitude.get_state().value = np.array([[args.amplitude]])
    sine = Sine()
    sine.hertz = hertz
    gain = Gain()
    gain.left = sine
    gain.right = amplitude
    sink = BlockDriver(rate=48000, blocksize=256)
    sink.input = gain

    samplerate, start_idx, worst = 48000, 0, 0.0
    while sink.tell() < args.blocks:
        outdata = sink.pull()
        t = ((start_idx + np.arange(len(outdata))) / samplerate).reshape(-1, 1)
        worst = max(worst, float(np.max(np.abs(outdata - args.amplitude * np.sin(2 * np.pi * args.frequency * t)))))
        start_idx += len(outdata)
    print(f'{args.blocks} blocks of 256 frames, {args.frequency} Hz x {args.amplitude}: max |err| vs formula = {worst:.3e}')
    return worst


if __name__ == '__main__':
    main()
