from .sequence import Sequence, SequenceStatus
from .block_manager import BlockManager, Block
from .scheduler import Scheduler
