"""The sequential twin of the kernel a program's FIRST launch would run (capi_ladder.hip scanDeviceOnce's routing, as scripts/cpu_fuzz.py and
scripts/explain.py follow it): rows of FindAllIndex on the CPU through the device images, without a GPU.  Test infrastructure."""
import struct

import numpy as np

import emu


def rows_on_twin(rx, hay):
    """(M, 2) int64 rows, or an int < 0 when the twin raises the kernel's fallback flag (the device would take the next rung)."""
    a = np.frombuffer(bytes(hay), dtype=np.uint8) if not isinstance(hay, np.ndarray) else hay
    blob = rx.blob()
    kind, flags = struct.unpack_from("<II", blob, 4)
    if rx.nullable == 2:                                             # every match is empty: no device program
        return emu.merge_empty_matches(np.zeros((0, 2), dtype=np.int64), a.size)
    if rx.delimiters is not None:
        got = emu.find_all_delim(*rx.delimiters, a)
    elif kind == 5:                                                  # transducer only
        got = emu.find_all_fsm(rx.fsm_image(), a, 3840, 32)
        if isinstance(got, int) and got in (-18, -32):
            got = emu.find_all_fsm(rx.fsm_image(), a, 3840, 32, dense=1)
    elif kind == 3:
        got = emu.find_all_charclass_wave(blob, a) if flags & 64 else emu.find_all(blob, a)
    elif kind == 4 or flags & 256:                                   # literal set / literal prefixes + anchored DFA
        got = emu.find_all_teddy_wave(blob, a)
    elif flags & 16:                                                 # complete ordered chain
        got = emu.find_all_chain6_bounded(blob, rx.chain_bounds()[0], a, 3840, 256) if flags & 512 else emu.find_all_chain6(blob, a, 3840, 256)
    elif rx.fsm_image() is not None:                                 # the transducer in front of the table-walking kernels
        got = emu.find_all_fsm(rx.fsm_image(), a, 3840, 32)
        if isinstance(got, int) and got in (-18, -32):
            got = emu.find_all_fsm(rx.fsm_image(), a, 3840, 32, dense=1)
        if isinstance(got, int):                                     # denser still (`.`: a match per byte): the table-walking kernel, the ladder's last rung
            got = emu.find_all(blob, a)
    else:
        got = emu.find_all(blob, a)
    if got is None:
        return -1
    if rx.nullable and not isinstance(got, int):
        got = emu.merge_empty_matches(got, a.size)
    return got
