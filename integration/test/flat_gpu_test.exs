defmodule Vettore.Index.FlatGpuTest do
  @moduledoc """
  `index: Vettore.Index.FlatGpu` plugged into Vettore's own entry points, on the scenarios the
  repository's parity suite runs through its Python mirror: `tests/golden/elixir_nif.json` holds them
  as data (inputs and expected outputs transcribed from the reference's tests, each with its `cite`),
  and every collection here is made twice -- with the built-in flat index and with the GPU one -- so
  that besides the fixture's expectations the two must agree on ids, order and raw scores.

  Needs an MI355X (the library has no CPU fallback: without a device `Vettore.new/1` returns
  `{:error, "no HIP device visible ..."}` and these tests say so).
  """
  use ExUnit.Case, async: false

  @fixture Path.expand("../../tests/golden/elixir_nif.json", __DIR__)
  @cases @fixture |> File.read!() |> JSON.decode!()

  defp collection(metric, dims, extra \\ []) do
    opts = [dimensions: dims, metric: String.to_atom(metric)] ++ extra
    {:ok, cpu} = Vettore.new(opts)
    {:ok, gpu} = Vettore.new(opts ++ [index: Vettore.Index.FlatGpu, index_options: [device: 0]])
    {cpu, gpu}
  end

  defp put_rows(collections, rows) do
    for c <- Tuple.to_list(collections) do
      :ok = Vettore.put_many(c, Enum.map(rows, fn [id, v] -> %{id: id, value: id, vector: v} end))
    end
  end

  defp both(collections, fun) do
    {cpu, gpu} = collections
    {fun.(cpu), fun.(gpu)}
  end

  defp triples({:ok, results}), do: Enum.map(results, &{&1.id, &1.score, &1.distance})

  test "all supported metrics return stable top-k results" do
    c = @cases["all_supported_metrics_return_stable_top_k_results"]

    for metric <- c["metrics"] do
      cs = collection(metric, 2)
      put_rows(cs, c["rows"])
      {want, got} = both(cs, &Vettore.search(&1, c["query"], limit: c["limit"]))
      assert Enum.map(elem(got, 1), & &1.id) == c["expect_ids"], metric
      assert triples(got) == triples(want), metric
    end
  end

  test "a native id that is not in ETS is dropped; Ok(()) arrives as {:ok, {}}" do
    c = @cases["phantom_native_id_and_ok_unit"]
    {:ok, gpu} = Vettore.new(dimensions: 1, metric: :l2, index: Vettore.Index.FlatGpu)
    [id, vector] = c["flat_insert"]
    assert Vettore.Gpu.Nifs.flat_insert(gpu.index_state, id, vector) == {:ok, {}}
    assert Vettore.search(gpu, [0.0], limit: 1) == {:ok, []}
    assert Vettore.Gpu.Nifs.flat_insert(gpu.index_state, "e", []) == {:error, c["put_empty_error"]}
  end

  test "stateless helpers: vector_top_k under every metric code, binary_top_k" do
    c = @cases["batched_native_helpers"]
    vectors = Enum.map(c["vectors"], fn [id, v] -> {id, v} end)

    for code <- c["metric_codes"] do
      {:ok, hits} = Vettore.Gpu.Nifs.vector_top_k(vectors, c["query"], code, c["dimensions"], c["limit"])
      assert Enum.map(hits, &elem(&1, 0)) == c["expect_ids"]
      assert {:ok, hits} == Vettore.Nifs.vector_top_k(vectors, c["query"], code, c["dimensions"], c["limit"])
    end

    [bad_code, message] = c["unknown_metric"]
    assert Vettore.Gpu.Nifs.vector_top_k(vectors, c["query"], bad_code, c["dimensions"], c["limit"]) == {:error, message}
    [code, dims, message] = c["bad_prefix"]
    assert Vettore.Gpu.Nifs.vector_top_k(vectors, c["query"], code, dims, c["limit"]) == {:error, message}
    b = c["binary"]
    words = Enum.map(b["vectors"], fn [id, w] -> {id, w} end)

    assert Vettore.Gpu.Nifs.binary_top_k(words, b["query"], b["dimensions"], b["limit"]) ==
             {:ok, Enum.map(b["expect"], fn [id, d] -> {id, d} end)}
  end

  test "cosine collection: score and distance of the first hit" do
    c = @cases["cosine_collection_result_semantics"]
    cs = collection(c["metric"], 2, normalize: String.to_atom(c["normalize"]), score: String.to_atom(c["score"]))
    put_rows(cs, c["rows"])
    {want, got} = both(cs, &Vettore.search(&1, c["query"], limit: c["limit"]))
    {:ok, [first | _]} = got
    e = c["expect_first"]
    assert {first.id, first.score, first.distance} == {e["id"], e["score"], e["distance"]}
    assert triples(got) == triples(want)
  end

  test "quantized search on the resident corpus equals the collection's own" do
    c = @cases["binary_quantized_search"]
    cs = collection(c["metric"], 2)
    put_rows(cs, c["rows"])
    {cpu, gpu} = cs
    opts = [candidates: c["candidates"], limit: c["limit"]]
    want = Vettore.quantized_search(cpu, c["query"], opts)
    got = Vettore.Index.FlatGpu.quantized_search(gpu, c["query"], opts)
    assert triples(got) == triples(want)
    [e] = c["expect"]
    {:ok, [hit]} = got
    assert {hit.id, hit.distance} == {e["id"], e["distance"]}
    # and the entry point of the collection itself keeps working on a GPU-indexed collection (it reads ETS)
    assert triples(Vettore.quantized_search(gpu, c["query"], opts)) == triples(want)
  end

  test "funnel / quantized with every row as a candidate agree with the exact flat search" do
    c = @cases["full_candidate_adaptive_modes_agree_with_exact_flat_search"]
    cs = collection(c["metric"], length(hd(c["rows"]) |> Enum.at(1)))
    put_rows(cs, c["rows"])
    {cpu, gpu} = cs
    n = length(c["rows"])
    query = c["query"]
    {:ok, exact} = Vettore.search(cpu, query, limit: 10)
    ids = Enum.map(exact, & &1.id)
    assert {:ok, on_gpu} = Vettore.search(gpu, query, limit: 10)
    assert triples({:ok, on_gpu}) == triples({:ok, exact})
    assert {:ok, q} = Vettore.Index.FlatGpu.quantized_search(gpu, query, candidates: n, limit: 10)
    assert Enum.map(q, & &1.id) == ids
    assert {:ok, f} = Vettore.Index.FlatGpu.funnel_search(gpu, query, candidates: n, limit: 10, stages: [2])
    assert Enum.map(f, & &1.id) == ids
    assert {:ok, [b]} = Vettore.Index.FlatGpu.search_batch(gpu, [query], limit: 10)
    assert triples({:ok, b}) == triples({:ok, exact})
  end

  test "hybrid search and the batched staged searches on the resident corpus" do
    c = @cases["full_candidate_adaptive_modes_agree_with_exact_flat_search"]
    cs = collection(c["metric"], length(hd(c["rows"]) |> Enum.at(1)))
    put_rows(cs, c["rows"])
    {cpu, gpu} = cs
    query = c["query"]
    n = c["candidates"]
    {:ok, exact} = Vettore.search(cpu, query, limit: c["limit"])
    ids = Enum.map(exact, & &1.id)
    generators = [funnel: [stages: [2, 4], candidates: n], quantized: [candidates: n], search: [candidates: n]]
    # the collection's own hybrid search (ETS + CPU NIFs) and the adapter's (one call on the resident rows) agree
    want = Vettore.hybrid_search(cpu, query, generators: generators, limit: c["limit"])
    got = Vettore.Index.FlatGpu.hybrid_search(gpu, query, generators: generators, limit: c["limit"])
    assert triples(got) == triples(want)
    assert Enum.map(elem(got, 1), & &1.id) == ids
    # default generators ([:funnel, :quantized], collection.ex:512-513) and a bare atom
    assert triples(Vettore.Index.FlatGpu.hybrid_search(gpu, query, limit: 3)) == triples(Vettore.hybrid_search(cpu, query, limit: 3))
    assert {:ok, [_ | _]} = Vettore.Index.FlatGpu.hybrid_search(gpu, query, generators: [:search], limit: 3)
    # errors as run_hybrid_generator's (collection.ex:536-556, :1136-1142)
    assert Vettore.Index.FlatGpu.hybrid_search(gpu, query, generators: []) == {:error, :invalid_generators}
    assert Vettore.Index.FlatGpu.hybrid_search(gpu, query, generators: [:nope]) == {:error, {:unknown_generator, :nope}}
    assert Vettore.Index.FlatGpu.hybrid_search(gpu, query, generators: [:hnsw]) == {:error, :hnsw_index_required}
    assert Vettore.Index.FlatGpu.hybrid_search(gpu, query, generators: [funnel: [stages: [5]]]) == {:error, :invalid_stages}
    assert Vettore.Index.FlatGpu.hybrid_search(gpu, query, generators: [quantized: [stages: [2]]]) == {:error, {:unsupported_option, :stages}}
    assert Vettore.Index.FlatGpu.hybrid_search(gpu, query, rerank: :nope) == {:error, {:invalid_rerank, :nope}}
    # batches: every query's list equals its own call
    queries = [query, Enum.map(query, &(&1 / 2)), [0.0, 0.0, 0.0, 0.0]]
    {:ok, qb} = Vettore.Index.FlatGpu.quantized_search_batch(gpu, queries, candidates: n, limit: c["limit"])
    {:ok, fb} = Vettore.Index.FlatGpu.funnel_search_batch(gpu, queries, stages: [2, 4], candidates: n, limit: c["limit"])
    for {q, i} <- Enum.with_index(queries) do
      assert triples({:ok, Enum.at(qb, i)}) == triples(Vettore.Index.FlatGpu.quantized_search(gpu, q, candidates: n, limit: c["limit"]))
      assert triples({:ok, Enum.at(fb, i)}) == triples(Vettore.Index.FlatGpu.funnel_search(gpu, q, stages: [2, 4], candidates: n, limit: c["limit"]))
    end
    assert Enum.map(hd(qb), & &1.id) == ids and Enum.map(hd(fb), & &1.id) == ids
    assert Vettore.Index.FlatGpu.quantized_search_batch(gpu, queries, candidates: 3, limit: 10) == {:error, :invalid_candidates}
    assert Vettore.Index.FlatGpu.funnel_search_batch(gpu, [[1.0]], limit: 1) == {:error, :dimension_mismatch}
  end

  test "adapter validation: limits and query length" do
    c = @cases["adapter_validation"]
    {:ok, gpu} = Vettore.new(dimensions: c["dimensions"], metric: String.to_atom(c["metric"]), index: Vettore.Index.FlatGpu)
    :ok = Vettore.put(gpu, %{id: "a", value: "a", vector: [0.0, 1.0]})
    for limit <- c["invalid_limits"], do: assert(Vettore.search(gpu, [0.0, 1.0], limit: limit) == {:error, :invalid_limit})
    assert Vettore.search(gpu, c["dimension_mismatch_query"], limit: 1) == {:error, :dimension_mismatch}
  end

  test "mutations follow the built-in index: upsert, delete, an emptied index takes another dimension" do
    cs = collection("l2", 2)
    put_rows(cs, [["b", [2.0, 0.0]], ["a", [0.0, 0.0]], ["c", [2.0, 0.0]]])
    agree = fn q, k -> {w, g} = both(cs, &Vettore.search(&1, q, limit: k)); assert triples(g) == triples(w); g end
    assert {:ok, [%{id: "a"}, %{id: "b"}]} = agree.([1.0, 0.0], 2)
    for c <- Tuple.to_list(cs), do: :ok = Vettore.put(c, %{id: "a", value: "a", vector: [10.0, 0.0]})
    assert {:ok, [%{id: "b"}]} = agree.([2.0, 0.0], 1)
    for c <- Tuple.to_list(cs), do: :ok = Vettore.delete(c, "b")
    assert {:ok, [%{id: "c"}]} = agree.([2.0, 0.0], 1)
  end

  test "one resource over several shards gives the same hits (shards may share a device)" do
    rows = for i <- 0..199, do: ["doc-#{i}", [:math.sin(i), :math.cos(3 * i), rem(i, 7) / 7]]
    {:ok, one} = Vettore.new(dimensions: 3, metric: :cosine, index: Vettore.Index.FlatGpu)
    {:ok, four} = Vettore.new(dimensions: 3, metric: :cosine, index: Vettore.Index.FlatGpu, index_options: [devices: [0, 0, 0, 0]])
    put_rows({one, four}, rows)
    assert triples(Vettore.search(four, [0.3, -0.2, 0.9], limit: 25)) == triples(Vettore.search(one, [0.3, -0.2, 0.9], limit: 25))
  end
end
