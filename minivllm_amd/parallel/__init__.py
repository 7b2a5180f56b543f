from .group import (init_tensor_parallel, destroy_tensor_parallel, set_tp_for_testing, tp_rank,
                    tp_size, tp_group, tp_all_reduce, tp_all_gather, tp_all_gather_stack, tp_gather, host_staged_collectives, tp_barrier,
                    env_rank_world, is_initialized)
