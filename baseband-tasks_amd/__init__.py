"""baseband_tasks_amd: MI355X-native coherent dedispersion and channelization
behind the stream-reader interface of mhvk/baseband-tasks.

The dedispersion -> channelizer hot path and its immediate neighbours
(detection, integration) are provided;
every task here runs hand-written gfx950 kernels through libbbt_hip.so and
raises if that library is missing -- there is no CPU fallback.
"""
from . import units
from .units import Time
from .base import (Base, BaseTaskBase, TaskBase, PaddedTaskBase, Task, SetAttribute, SinglePrecision)
from .generators import (StreamGenerator, EmptyStreamGenerator, Noise, NoiseGenerator,
                         DeviceStream, HostStream)
from .dm import DispersionMeasure
from .fourier import fft_maker, HipFFTMaker
from .dispersion import Disperse, Dedisperse, DisperseSamples, DedisperseSamples
from .convolution import Convolve, ConvolveSamples
from .sampling import ShiftAndResample, Resample, TimeDelay, ShiftSamples
from .channelize import Channelize, Dechannelize
from .pfb import (sinc_hamming, PolyphaseFilterBank, PolyphaseFilterBankSamples,
                  InversePolyphaseFilterBank)
from .functions import Square, Power
from .integration import Integrate
from .ingest import RawFrameStream, open_vdif, open_dada
from . import hip
from . import hdf5

__version__ = '0.1.0'
