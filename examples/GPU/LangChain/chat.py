"""LangChain LLM + embeddings backed by low-bit models on the GPU
(reference: python/llm/example/GPU/LangChain/{chat.py,rag.py}, ipex_llm/langchain/llms/transformersllm.py:87-225)."""
import argparse

from ipex_llm.langchain.embeddings import TransformersEmbeddings
from ipex_llm.langchain.llms import TransformersLLM

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-path", required=True)
    ap.add_argument("--question", default="What is AI?")
    a = ap.parse_args()
    llm = TransformersLLM.from_model_id(model_id=a.model_path, model_kwargs={"temperature": 0, "max_length": 256,
                                                                             "trust_remote_code": True}, device_map="cuda")
    print(llm.invoke(f"USER: {a.question}\nASSISTANT:") if hasattr(llm, "invoke") else llm(f"USER: {a.question}\nASSISTANT:"))
    emb = TransformersEmbeddings.from_model_id(model_id=a.model_path, device_map="cuda")
    docs = ["MI355X has 288 GB of HBM3E.", "xGMI links are point to point.", "Paris is the capital of France."]
    import torch
    q, d = torch.tensor(emb.embed_query(a.question)), torch.tensor(emb.embed_documents(docs))
    print("closest document:", docs[int(torch.nn.functional.cosine_similarity(q[None], d).argmax())])
