"""AutoGen agents on a local low-bit model (counterpart of the reference's
example/GPU/Applications/autogen/teachability_new_knowledge.py).  The agents talk to the OpenAI-compatible endpoint of
this framework's serving stack, so nothing in AutoGen has to know about the backend:

    # 1. serve a model (either front end speaks /v1/chat/completions)
    python -m bigdl_amd.serving.fastapi.api_server --repo-id-or-model-path /models/Mistral-7B-Instruct-v0.2 \
        --low-bit sym_int4 --port 8000
    #    or: FastChat controller + python -m bigdl_amd.serving.fastchat.ipex_llm_worker ... + openai_api_server
    # 2. run the agents (needs `pyautogen` with the teachable extra)
    python teachability_new_knowledge.py --base-url http://127.0.0.1:8000/v1 --model Mistral-7B-Instruct-v0.2
"""
import argparse


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base-url", default="http://127.0.0.1:8000/v1")
    ap.add_argument("--model", default="local-model")
    a = ap.parse_args()
    from autogen import ConversableAgent, UserProxyAgent
    from autogen.agentchat.contrib.capabilities.teachability import Teachability
    llm_config = {"config_list": [{"model": a.model, "base_url": a.base_url, "api_key": "NULL"}], "timeout": 600,
                  "cache_seed": None, "max_tokens": 512}
    agent = ConversableAgent(name="teachable_agent", llm_config=llm_config)
    Teachability(verbosity=0, reset_db=True, path_to_db_dir="./tmp/teachability_db", recall_threshold=1.5).add_to_agent(agent)
    user = UserProxyAgent(name="user", human_input_mode="NEVER", is_termination_msg=lambda m: "TERMINATE" in (m.get("content") or ""),
                          max_consecutive_auto_reply=0, code_execution_config={"use_docker": False})
    user.initiate_chat(agent, message="What is the Vicuna model?", clear_history=True)
    user.initiate_chat(agent, message="Vicuna is a 13B-parameter language model released by Meta.", clear_history=True)
    user.initiate_chat(agent, message="What is the Orca model?", clear_history=True)
    user.initiate_chat(agent, message="Orca is a 13B-parameter language model developed by Microsoft. It outperforms Vicuna "
                                      "on most tasks.", clear_history=True)
    user.initiate_chat(agent, message="How does the Vicuna model compare to the Orca model?", clear_history=True)


if __name__ == "__main__":
    main()
