from .common import *  # noqa: F401,F403
from .common import Activation, Dense, LayerDictionary, get_activation  # noqa: F401
from .aggregators import BaseAggregator, GCNAggregator, MultiLinkGCNAggregator  # noqa: F401
from .layers import HeterGCNLayer, InnerProductLayer, StackedHeterGCNLayers  # noqa: F401
