"""Job body for tests/test_dp_engine.py: started by torch.distributed.run with one rank per
data-parallel replica (CPU, gloo).  Rank 0 passes the prompts and writes the merged result."""
import json
import sys

from test_dp_engine import make_cfg, make_prompts

from minivllm_amd.engine.dp_engine import DataParallelEngine
from minivllm_amd.sampling_parameters import SamplingParams


def main(out_path: str) -> None:
    eng = DataParallelEngine(make_cfg())
    prompts = make_prompts() if eng.rank == 0 else None
    res = eng.generate(prompts, SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))
    if eng.rank == 0:
        with open(out_path, "w") as f:
            json.dump(res["token_ids"], f)
    else:
        assert res is None
    eng.exit()


if __name__ == "__main__":
    main(sys.argv[1])
