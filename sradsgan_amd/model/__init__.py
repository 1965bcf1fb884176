from .sradsgan import (CGAM, CLAM, GAB_UP, MSB, RAB, SGAM, SLAM, Discriminator, FeatureExtractor,  # noqa: F401
                       GANLoss, GeneratorResNet, ResGroup)
from .base_networks import ChannelAttention, SpatialAttention  # noqa: F401
