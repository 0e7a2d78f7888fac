"""go_with_the_flows_amd -- MI355X-native discrete point-flow decoder (gfx950 HIP behind the reference's nn.Module API)."""
from .layers import SharedDot, Swish
from .flows import CondRealNVPFlow3D, CondRealNVPFlow3DTriple, WARP_PATTERNS
from .decoders import LocalCondRNVPDecoder
from .mixture import MixtureStack, flow_mixture_nll
from . import optim
from . import metrics
from . import evaluation
from .encoders import PointNetCloudEncoder, FeatureEncoder, WeightsEncoder
from .prior import RealNVPFlow, RealNVPFlowCouple, GlobalRNVPDecoder, GaussianFlowNLL, GaussianEntropy
from .models import Local_Cond_RNVP_MC_Global_RNVP_VAE, Flow_Mixture_Model, Flow_Mixture_Loss, FlowMixtureNLL
from ._lib import GwtfError

__all__ = ['SharedDot', 'Swish', 'CondRealNVPFlow3D', 'CondRealNVPFlow3DTriple', 'LocalCondRNVPDecoder',
           'WARP_PATTERNS', 'GwtfError', 'MixtureStack', 'flow_mixture_nll', 'optim', 'metrics', 'evaluation',
           'PointNetCloudEncoder', 'FeatureEncoder', 'WeightsEncoder', 'RealNVPFlow', 'RealNVPFlowCouple',
           'GlobalRNVPDecoder', 'GaussianFlowNLL', 'GaussianEntropy', 'Local_Cond_RNVP_MC_Global_RNVP_VAE',
           'Flow_Mixture_Model', 'Flow_Mixture_Loss', 'FlowMixtureNLL']
