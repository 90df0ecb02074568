def run(main):
    main([])
