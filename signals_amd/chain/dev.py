"""The name `signals.chain.dev.SinkDevice`, so that graph scripts written for the reference import and wire up
(`sink = SinkDevice(...); sink.input = node`).  Audio devices are out of scope: there is no stream, no
callback thread and no device list here.  The sink is the headless `BlockDriver` -- the caller pulls
blocks itself with `pull()` / `render()`; whatever device record a script passes is kept as `info` and otherwise ignored."""
from signals_amd.chain.driver import BlockDriver


class SinkDevice(BlockDriver):

    def __init__(self, info=None, rate: int = 48000, blocksize: int = 256):
        super().__init__(rate=int(getattr(info, 'default_samplerate', rate)), blocksize=blocksize)
        self.info = info
