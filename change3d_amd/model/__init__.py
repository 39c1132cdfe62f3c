from .change_decoder import ChangeDecoder  # noqa: F401
from .trainer import Encoder, Trainer  # noqa: F401
from .utils import BCEDiceLoss, FusedAdam, ParamArena, adjust_learning_rate, weight_init  # noqa: F401
from .x3d import create_x3d  # noqa: F401
